#!/usr/bin/env python
"""Synthesis driver with the reference's command line (reference predict_mel.py:2-15,58-74).

Usage: predict_mel.py --source-data-root=<dir> --target-data-root=<dir> --checkpoint-dir=<dir> --output-dir=<dir>
                      --selected-list-dir=<dir> [--checkpoint=<file>] [--selected-list-filename=<name>]
                      [--hparams=<a=b>] [--hparam-json-file=<path>]

For every key of the list: free-running decode (batch size 1) from `model-<step>.pt`, output `<key>.mfbsp` (raw
little-endian float32 [T, num_mels], post-net output when `use_postnet_v2`), `<key>.alignment.npz` + `<key>.png` (the
two alignment histories laid out [T_memory, T_query] as in the reference's predictions) and `<key>.tfrecord` (the
reference's prediction record, utils/tfrecord.py:135-152).  `use_forced_alignment_mode=True` re-decodes with both
attention mechanisms pinned to the alignments of the first pass (models/models.py:411-428).
Usage: see --help"""
import argparse
import glob
import os
import sys

import numpy as np
import torch


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
    from satt_amd.datasets.ljspeech import decode_source_record, decode_target_record
    from satt_amd.engine import Engine
    from satt_amd.hparams import hparams
    from satt_amd.inference import infer
    from satt_amd.params import ModelConfig
    from satt_amd.utils import tfrecord
    from satt_amd.utils.summary import plot_alignments
    from train import load_key_list

    if a.hparam_json_file:
        hparams.parse_json(open(a.hparam_json_file).read())
    hparams.parse(a.hparams)
    ck = a.checkpoint or max(glob.glob(os.path.join(a.checkpoint_dir, "model-*.pt")),
                             key=lambda p: int(p.rsplit("-", 1)[1][:-3]))
    state = torch.load(ck, map_location="cpu")
    eng = Engine(ModelConfig.from_hparams(hparams), "cuda")
    eng.flat.copy_(state["params"])
    for k, (m, v) in state["bn"].items():
        eng.bn[k][0].copy_(m); eng.bn[k][1].copy_(v)
    eng.refresh_shadows()
    os.makedirs(a.output_dir, exist_ok=True)
    for key in load_key_list(a.selected_list_filename, a.selected_list_dir):
        f = os.path.join(a.source_data_root, "%s.%s" % (key, hparams.source_file_extension))
        s = decode_source_record(next(tfrecord.read_records(f)))
        spk = np.array([s.speaker_id]) if s.speaker_id >= 0 else None
        out = infer(eng, s.source[None, :], np.array([s.source_length]), max_steps=hparams.max_iters, speaker_id=spk)
        if hparams.use_forced_alignment_mode:
            # second decode with both mechanisms pinned to the alignments just found (models/models.py:411-428)
            out = infer(eng, s.source[None, :], np.array([s.source_length]), max_steps=out["steps"], speaker_id=spk,
                        min_steps=1 << 30, teacher_alignments=(out["alignment1"], out["alignment2"]))
        mel = out["mel"][0].float().cpu().numpy().astype("<f4")
        assert mel.shape[1] == hparams.num_mels
        mel.tofile(os.path.join(a.output_dir, "%s.%s" % (key, hparams.predicted_mel_extension)))
        aligns = [out["alignment1"][0].cpu().numpy().T, out["alignment2"][0].cpu().numpy().T]     # [T_memory, T_query]
        np.savez(os.path.join(a.output_dir, "%s.alignment.npz" % key), alignment=aligns[0], alignment2=aligns[1])
        plot_alignments(os.path.join(a.output_dir, "%s.png" % key), aligns)
        gt = None
        if a.target_data_root:
            tf_ = os.path.join(a.target_data_root, "%s.%s" % (key, hparams.target_file_extension))
            if os.path.exists(tf_):
                gt = decode_target_record(next(tfrecord.read_records(tf_)))["mel"]
        tfrecord.write_prediction_result(s.id, key, aligns, mel, gt, s.text or "", s.source, None,
                                         os.path.join(a.output_dir, "%s.tfrecord" % key))
        print("%s: %d frames" % (key, mel.shape[0]))


if __name__ == "__main__":
    main()
