"""Training- / evaluation-time result dumps: the role of the reference's MetricsSaver hook (models/models.py:499-508:
`MetricsSaver([alignment1, alignment2] + self_attention_alignment, global_step, mel_output, labels.mel, labels.target_length,
features.id, features.text, params.alignment_save_steps, mode, summary_writer, save_training_time_metrics=...,
keep_eval_results_max_epoch=...)`).  The hook class itself lives in the un-vendored tacotron2 package; its in-tree sibling
for the MGC / LF0 models (modules/metrics.py:96-150) shows the protocol this follows: when `(stale_global_step + 1) %
save_steps == 0` (or at step 0) the batch's ids, texts, alignments, predicted and ground-truth features are fetched and written
as `<mode>_result_step<step:09d>_<ids>.tfrecord` beside the event files, plus one alignment plot per utterance.

Here: one prediction record per utterance (the reference's prediction-record feature names, utils/tfrecord.py:135-152) in
that file, and `alignment_step<step:09d>_<id>.png` (all attention histories stacked, memory axis up).  TRAIN mode dumps only
with `save_training_time_metrics` (hparams.py: default False - the hook then only serves evaluation), EVAL mode always; eval
results of more than `keep_eval_results_max_epoch` distinct steps ago are deleted."""
import glob
import os
import re

import numpy as np

from . import tfrecord
from .summary import plot_alignments


class MetricsSaver:
    def __init__(self, out_dir, save_steps, mode, save_training_time_metrics=False, keep_eval_results_max_epoch=10):
        self.out_dir, self.save_steps, self.mode = out_dir, max(1, int(save_steps)), mode
        self.enabled = mode == "eval" or bool(save_training_time_metrics)
        self.keep = int(keep_eval_results_max_epoch)

    def due(self, global_step_after):
        """global_step_after: the step counter AFTER the update (the reference tests stale_step + 1)"""
        return self.enabled and (global_step_after % self.save_steps == 0 or global_step_after == 1)

    def save(self, step, ids, keys, texts, sources, source_lengths, alignments, mel, gt_mel, target_lengths, r=1):
        """alignments: list of arrays [B, T_query, T_memory]; mel / gt_mel [B, Tm, num_mels]; writes one file per batch"""
        os.makedirs(self.out_dir, exist_ok=True)
        B = len(ids)
        name = "%s_result_step%09d_%s.tfrecord" % (self.mode, step, ",".join(str(int(i)) for i in ids))
        payloads, written = [], []
        for i in range(B):
            L, T = int(source_lengths[i]), int(target_lengths[i])
            al = [np.asarray(a[i])[:max(1, T // r), :L].T if a.shape[-1] >= L else np.asarray(a[i]).T for a in alignments]
            tmp = os.path.join(self.out_dir, ".tmp_pred.tfrecord")
            tfrecord.write_prediction_result(int(ids[i]), str(keys[i]), al, np.asarray(mel[i])[:T], np.asarray(gt_mel[i])[:T],
                                             str(texts[i]), np.asarray(sources[i])[:L], None, tmp)
            payloads.append(next(tfrecord.read_records(tmp)))
            os.remove(tmp)
            png = os.path.join(self.out_dir, "alignment_step%09d_%d.png" % (step, int(ids[i])))
            plot_alignments(png, al)
            written.append(png)
        path = os.path.join(self.out_dir, name)
        tfrecord.write_records(path, payloads)
        if self.mode == "eval" and self.keep > 0:
            self._prune()
        return [path] + written

    def _prune(self):
        pat = re.compile(r"(?:eval_result|alignment)_step(\d{9})_")
        files = [(int(m.group(1)), p) for p in glob.glob(os.path.join(self.out_dir, "*")) for m in [pat.search(os.path.basename(p))] if m]
        steps = sorted({s for s, _ in files})
        for s, p in files:
            if s in steps[:-self.keep]:
                os.remove(p)
