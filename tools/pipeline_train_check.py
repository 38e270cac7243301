"""Does the input pipeline keep the GPU busy?  (run on the GPU box: python tools/pipeline_train_check.py)

Writes N LJSpeech-sized utterances in the reference's on-disk layout (one `<key>.source.tfrecord` + `<key>.target.tfrecord`
per utterance, reference datasets/ljspeech/dataset.py:52-72) to tmpfs, then measures
  (1) the pipeline alone: utterances/s of create_from_tfrecord_files(...).shuffle().group_by_batch().prefetch(pin_memory=True);
  (2) train steps on the SAME batches already resident on the device (the GPU-bound time);
  (3) model.train(...) reading through the pipeline (what train.py does), wall clock per step.
(3) <= (2) + a few percent means the reader is off the critical path."""
import json
import os
import shutil
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import satt_amd  # noqa: F401
    from satt_amd.datasets import ljspeech
    from satt_amd.hparams import hparams as default_hparams
    from satt_amd.models.models import RunConfig, tacotron_model_factory
    from satt_amd.utils import tfrecord
    N = int(os.environ.get("N_UTT", "512"))
    steps = int(os.environ.get("STEPS", "48"))
    base = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
    d = os.path.join(base, "satt_pipeline_check")
    shutil.rmtree(d, ignore_errors=True); os.makedirs(d)
    g = np.random.default_rng(0)
    src, tgt = [], []
    for i in range(N):
        L, T = int(g.integers(60, 161)), int(g.integers(300, 796))
        key = ("LJ%05d" % i).encode()
        ps, pt = os.path.join(d, "LJ%05d.source.tfrecord" % i), os.path.join(d, "LJ%05d.target.tfrecord" % i)
        tfrecord.write_records(ps, [tfrecord.make_example({"id": i, "key": key, "source": np.concatenate(
            [[0], g.integers(1, 60, L - 2), [0]]).astype("<i8").tobytes(), "source_length": L, "text": b"synthetic"})])
        tfrecord.write_records(pt, [tfrecord.make_example({"id": i, "key": key, "mel": g.normal(-40, 10, (T, 80)).astype("<f4").tobytes(),
                                                           "mel_width": 80, "target_length": T})])
        src.append(ps); tgt.append(pt)
    hp = default_hparams.copy()
    dd = json.load(open(os.path.join(ROOT, "examples", "ljspeech", "self-attention-tacotron.json"))); dd.pop("_comment", None)
    hp.parse_json(json.dumps(dd))
    hp.average_mel_level_db = [-40.0] * 80
    hp.stddev_mel_level_db = [10.0] * 80
    hp.parse("log_step_count_steps=100000,save_checkpoints_steps=100000,alignment_save_steps=100000")
    par = ljspeech.get_parallelism(hp.interleave_cycle_length_cpu_factor, hp.interleave_cycle_length_min, hp.interleave_cycle_length_max)

    def pipeline(pin=True):
        return ljspeech.create_from_tfrecord_files(src, tgt, hp, cycle_length=par).prepare_and_zip().filter_by_max_output_length() \
            .repeat().shuffle(hp.suffle_buffer_size, seed=1).group_by_batch().prefetch(hp.prefetch_buffer_size, pin_memory=pin)
    print("host: %d cores, interleave parallelism %d, %d utterances (%.0f MB) in %s" %
          (os.cpu_count(), par, N, sum(os.path.getsize(p) for p in tgt) / 1e6, d))
    # (1) the pipeline alone
    for pin in (False, True):
        it = iter(pipeline(pin))
        next(it)
        t0 = time.perf_counter(); n = 0
        for _ in range(steps):
            n += next(it)["mel"].shape[0]
        dt = time.perf_counter() - t0
        print("pipeline alone (%s batch memory): %.0f utterances/s, %.2f ms per batch of %d"
              % ("page-locked ring" if pin else "fresh pageable", n / dt, dt / steps * 1e3, hp.batch_size))
        it.close() if hasattr(it, "close") else None
    model = tacotron_model_factory(hp, None, RunConfig.from_hparams(hp), device="cuda", rng_seed=0)
    eng = model.engine
    # (2) the same batches, resident
    it = iter(pipeline(True))
    host = []
    for _ in range(steps + 8):
        b = next(it)
        host.append({k: (np.array(v) if isinstance(v, np.ndarray) else v) for k, v in b.items()})
    it.close() if hasattr(it, "close") else None
    tp = torch.as_tensor(next(iter(pipeline(True)))["mel"])
    print("batch tensors of the pinned pipeline report is_pinned() = %s" % tp.is_pinned())
    dev = [eng.to_device_batch({k: v for k, v in b.items() if isinstance(v, np.ndarray) and k != "id"}) for b in host]
    for b in dev[:8]:
        eng.train_step(b); eng.optimizer_step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for b in dev[8:]:
        ctx = eng.train_step(b); eng.optimizer_step()
    torch.cuda.synchronize()
    res = (time.perf_counter() - t0) / steps * 1e3
    eng.check_clusters(ctx)
    shapes = sorted({(b["mel"].shape[1]) for b in host})
    print("resident batches (Tm %d..%d): %.2f ms/step" % (shapes[0], shapes[-1], res))
    # (3) through the pipeline, as train.py runs it
    for pin in (True, False):
        it = iter(pipeline(pin))               # ONE pipeline for warm-up and timed steps: the readers and the ring are started once
        model.train(it, steps=12)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        model.train(it, steps=steps)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        print("model.train through the pipeline (%s): %.2f ms/step (%.1f %% over resident)"
              % ("page-locked ring, async upload" if pin else "pageable, blocking upload", ms, (ms / res - 1) * 100))
        it.close() if hasattr(it, "close") else None
    shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
