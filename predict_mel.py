#!/usr/bin/env python
"""Synthesis driver with the reference's command line (reference predict_mel.py:2-15,58-74).

Usage: predict_mel.py --source-data-root=<dir> --target-data-root=<dir> --checkpoint-dir=<dir> --output-dir=<dir>
                      --selected-list-dir=<dir> [--checkpoint=<file>] [--selected-list-filename=<name>]
                      [--hparams=<a=b>] [--hparam-json-file=<path>]

For every key of the list: free-running decode (batch size 1) from `model-<step>.pt`, output `<key>.mfbsp` (raw
little-endian float32 [T, num_mels]; the PostNetV2 output when `use_postnet_v2`, models/models.py:440-462), `<key>.alignment.npz` + `<key>.png` (the
two alignment histories laid out [T_memory, T_query] as in the reference's predictions) and `<key>.tfrecord` (the
reference's prediction record, utils/tfrecord.py:135-152).  `use_forced_alignment_mode=True` (models/models.py:387-428):
pass 1 is the validation decode fed with the GROUND-TRUTH mel (needs --target-data-root), pass 2 feeds its own outputs
back while both attention mechanisms return pass 1's alignments.
Usage: see --help"""
import argparse
import glob
import os
import sys

import numpy as np


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--source-data-root", required=True)
    ap.add_argument("--target-data-root", default=None)
    ap.add_argument("--checkpoint-dir", required=True)
    ap.add_argument("--output-dir", required=True)
    ap.add_argument("--selected-list-dir", required=True)
    ap.add_argument("--checkpoint", default=None)
    ap.add_argument("--selected-list-filename", default="test.csv")
    ap.add_argument("--hparams", default="")
    ap.add_argument("--hparam-json-file", default=None)
    a = ap.parse_args(argv)

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import satt_amd  # noqa: F401
    from satt_amd.datasets.dataset_factory import dataset_factory
    from satt_amd.hparams import hparams
    from satt_amd.models.models import tacotron_model_factory
    from satt_amd.utils import tfrecord
    from satt_amd.utils.summary import plot_alignments
    from train import load_key_list

    if a.hparam_json_file:
        hparams.parse_json(open(a.hparam_json_file).read())
    hparams.parse(a.hparams)
    if hparams.use_forced_alignment_mode and not a.target_data_root:
        raise SystemExit("use_forced_alignment_mode aligns to the reference audio: --target-data-root is required")
    # reference predict_mel.py:40-57: estimator = tacotron_model_factory(hparams, checkpoint_dir, run_config);
    # estimator.predict(input_fn, checkpoint_path=...)
    model = tacotron_model_factory(hparams, None, device="cuda")
    ck = a.checkpoint or max(glob.glob(os.path.join(a.checkpoint_dir, "model-*.pt")),
                             key=lambda p: int(p.rsplit("-", 1)[1][:-3]))
    model.restore(ck)
    os.makedirs(a.output_dir, exist_ok=True)
    keys = load_key_list(a.selected_list_filename, a.selected_list_dir)
    src = [os.path.join(a.source_data_root, "%s.%s" % (k, hparams.source_file_extension)) for k in keys]
    have_target = bool(a.target_data_root) and all(
        os.path.exists(os.path.join(a.target_data_root, "%s.%s" % (k, hparams.target_file_extension))) for k in keys)

    def input_fn():
        if have_target:
            tgt = [os.path.join(a.target_data_root, "%s.%s" % (k, hparams.target_file_extension)) for k in keys]
            # batch size 1, targets merged into the source side (reference predict_mel.py:46-50)
            return dataset_factory(src, tgt, hparams).prepare_and_zip().group_by_batch(batch_size=1) \
                .merge_target_to_source()
        from satt_amd.datasets.ljspeech import decode_source_record

        def gen():
            for f in src:
                s = decode_source_record(next(tfrecord.read_records(f)))
                b = dict(source=s.source[None, :], source_length=np.array([s.source_length], np.int64),
                         id=np.array([s.id], np.int64), key=[s.key], text=[s.text])
                if s.speaker_id >= 0:
                    b["speaker_id"] = np.array([s.speaker_id], np.int64)
                yield b
        return gen()

    for p in model.predict(input_fn):
        key = p["key"]
        mel = (p["mel_postnet"] if "mel_postnet" in p else p["mel"]).astype("<f4")
        assert mel.shape[1] == hparams.num_mels
        mel.tofile(os.path.join(a.output_dir, "%s.%s" % (key, hparams.predicted_mel_extension)))
        # [T_memory, T_query]; the single-source baseline model has one history (reference models/models.py:196-212)
        aligns = [p[k] for k in ("alignment", "alignment2") if k in p]
        np.savez(os.path.join(a.output_dir, "%s.alignment.npz" % key), **{k: p[k] for k in ("alignment", "alignment2") if k in p})
        plot_alignments(os.path.join(a.output_dir, "%s.png" % key), aligns)
        gt = None
        if a.target_data_root:          # the RAW reference mel (un-normalised), as the reference's record holds it
            from satt_amd.datasets.ljspeech import decode_target_record
            tf_ = os.path.join(a.target_data_root, "%s.%s" % (key, hparams.target_file_extension))
            if os.path.exists(tf_):
                gt = decode_target_record(next(tfrecord.read_records(tf_)))["mel"]
        tfrecord.write_prediction_result(p["id"], key, aligns, mel, gt, p["text"] or "", p["source"], None,
                                         os.path.join(a.output_dir, "%s.tfrecord" % key))
        print("%s: %d frames" % (key, mel.shape[0]))


if __name__ == "__main__":
    main()
