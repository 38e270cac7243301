"""reference datasets/vctk/dataset.py: same pipeline as LJSpeech; the records additionally carry speaker_id, age, gender
(:36-38,70-72), which `decode_source_record` reads when present and `pad_batch` forwards as `speaker_id`."""
from .ljspeech import BatchedDataset, DatasetSource, MelData, SourceData  # noqa: F401
