"""reference datasets/dataset_factory.py:12-35 under its own module name"""
from .ljspeech import create_from_tfrecord_files, dataset_factory  # noqa: F401
