#!/usr/bin/env python
"""Training driver with the reference's command line (reference train.py:7-18).

Usage: train.py --source-data-root=<dir> --target-data-root=<dir> --checkpoint-dir=<dir> --selected-list-dir=<dir>
                [--hparams=<a=b,c=d>] [--hparam-json-file=<path>] [--multi-gpus] [--max-steps=<n>]

Reads `<key>.source.tfrecord` / `<key>.target.tfrecord` of the keys in `train.csv` (TensorFlow-free reader), runs the
MI355X training engine, writes `model-<step>.pt` checkpoints every `save_checkpoints_steps` steps, the loss to
`hparams.logfile` and TensorBoard scalars (`events.out.tfevents.*`, reference names) to the checkpoint directory; with
a `validation.csv` in the list directory every checkpoint is followed by the EVAL double pass (free run + teacher-fed).
`--multi-gpus` is ONE command, as in the reference (train.py:16,68,74: MirroredStrategy over every visible GPU): with WORLD_SIZE unset
this process re-executes itself under torch.distributed.run with one rank per visible GPU (SATT_NUM_GPUS=N restricts it to N) -
one process per GPU, RCCL gradient all-reduce, every rank reads its own shard of the key list.  Under an external launcher
(`python -m torch.distributed.run --nproc-per-node N train.py ... --multi-gpus`) it runs as the rank it is given."""
import argparse
import logging
import os
import sys

import torch


def load_key_list(filename, in_dir):
    with open(os.path.join(in_dir, filename), mode="r", encoding="utf-8") as f:
        return [ln.strip() for ln in f if ln.strip()]


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--source-data-root", required=True)
    ap.add_argument("--target-data-root", required=True)
    ap.add_argument("--checkpoint-dir", required=True)
    ap.add_argument("--selected-list-dir", required=True)
    ap.add_argument("--hparams", default="")
    ap.add_argument("--hparam-json-file", default=None)
    ap.add_argument("--multi-gpus", action="store_true")
    ap.add_argument("--max-steps", type=int, default=None, help="stop after this many optimiser steps")
    a = ap.parse_args(argv)

    if a.multi_gpus and "WORLD_SIZE" not in os.environ:
        n = int(os.environ.get("SATT_NUM_GPUS", "0")) or torch.cuda.device_count()
        if n < 1:
            raise SystemExit("--multi-gpus: no GPU visible")
        if n > 1:          # become the launcher of the N ranks (the exit code is the job's)
            import socket
            import subprocess
            sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
            env = dict(os.environ)
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            env.setdefault("OMP_NUM_THREADS", "8")
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
                   "--master-port", str(port), os.path.abspath(__file__)] + list(sys.argv[1:] if argv is None else argv)
            raise SystemExit(subprocess.call(cmd, env=env))

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import satt_amd  # noqa: F401
    from satt_amd.datasets.dataset_factory import create_from_tfrecord_files, dataset_factory
    from satt_amd.datasets.ljspeech import get_parallelism
    from satt_amd.hparams import hparams
    from satt_amd.models.models import RunConfig, tacotron_model_factory
    from satt_amd.parallel import DataParallel
    from satt_amd.utils.summary import EventFileWriter

    if a.hparam_json_file:
        hparams.parse_json(open(a.hparam_json_file).read())
    hparams.parse(a.hparams)
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not a.multi_gpus:
        raise SystemExit("WORLD_SIZE > 1 needs --multi-gpus")
    if torch.cuda.device_count() < world:
        raise SystemExit("WORLD_SIZE=%d but only %d device(s) visible" % (world, torch.cuda.device_count()))
    torch.cuda.set_device(local)
    dp = DataParallel(world, rank, local)
    if dp.active and torch.distributed.get_world_size() != world:
        raise SystemExit("the process group has %d rank(s), WORLD_SIZE=%d" % (torch.distributed.get_world_size(), world))
    keys = load_key_list("train.csv", a.selected_list_dir)[rank::world]
    src = [os.path.join(a.source_data_root, "%s.%s" % (k, hparams.source_file_extension)) for k in keys]
    tgt = [os.path.join(a.target_data_root, "%s.%s" % (k, hparams.target_file_extension)) for k in keys]
    os.makedirs(a.checkpoint_dir, exist_ok=True)
    logging.basicConfig(level=logging.INFO, format="%(asctime)s - %(levelname)s - %(message)s",
                        handlers=[logging.StreamHandler()] + ([logging.FileHandler(hparams.logfile)] if rank == 0 else []))
    # reference train.py:60-98: RunConfig + tacotron_model_factory(hparams, model_dir, run_config) + estimator.train(...).
    # Unknown / unbuilt model, encoder, decoder or attention names raise here (never a silent substitute); an existing
    # model-<step>.pt in the checkpoint directory is resumed (parameters, Adam moments, BatchNorm statistics, step).
    model = tacotron_model_factory(hparams, a.checkpoint_dir, RunConfig.from_hparams(hparams),
                                   device="cuda:%d" % local, dp=dp)
    if model.global_step:
        logging.info("resumed from step %d", model.global_step)

    # reference train.py:34-36: reader parallelism of the interleave from the hparams and the host's core count
    interleave_parallelism = get_parallelism(hparams.interleave_cycle_length_cpu_factor, hparams.interleave_cycle_length_min,
                                             hparams.interleave_cycle_length_max)
    logging.info("Interleave parallelism is %d.", interleave_parallelism)

    def train_input_fn():
        # reference train.py:40-55.  A resumed run must not replay the data order of the first one: the shuffle seed moves with
        # the restored step.  Batches are assembled in page-locked memory (the H2D copy of the next batch is asynchronous).
        ds = create_from_tfrecord_files(src, tgt, hparams, cycle_length=interleave_parallelism,
                                        buffer_output_elements=hparams.interleave_buffer_output_elements,
                                        prefetch_input_elements=hparams.interleave_prefetch_input_elements).prepare_and_zip()
        ds = ds.cache(hparams.cache_file_name) if hparams.use_cache else ds
        return ds.filter_by_max_output_length().repeat(count=None) \
            .shuffle(hparams.suffle_buffer_size, seed=rank + 7919 * model.global_step).group_by_batch() \
            .prefetch(hparams.prefetch_buffer_size, pin_memory=True)
    # observability: TensorBoard event files in the checkpoint directory with the reference's scalar
    # names (models/models.py:600-616); EVAL double pass on validation.csv at every checkpoint (models/models.py:517-564)
    writer = EventFileWriter(a.checkpoint_dir) if rank == 0 else None
    eval_writer = EventFileWriter(os.path.join(a.checkpoint_dir, "eval")) if rank == 0 else None
    eval_input_fn = None
    if rank == 0 and os.path.exists(os.path.join(a.selected_list_dir, "validation.csv")):
        vkeys = load_key_list("validation.csv", a.selected_list_dir)[:hparams.num_evaluation_steps * hparams.batch_size]
        vsrc = [os.path.join(a.source_data_root, "%s.%s" % (k, hparams.source_file_extension)) for k in vkeys]
        vtgt = [os.path.join(a.target_data_root, "%s.%s" % (k, hparams.target_file_extension)) for k in vkeys]
        if vkeys:
            eval_input_fn = lambda: dataset_factory(vsrc, vtgt, hparams).prepare_and_zip() \
                .filter_by_max_output_length().group_by_batch()

    def on_checkpoint(step, path):
        if eval_input_fn is None:
            return
        ev = model.evaluate(eval_input_fn, steps=hparams.num_evaluation_steps)
        scalars = {k: v for k, v in ev.items() if k != "global_step"}
        if scalars:
            eval_writer.add_scalars(step, scalars); eval_writer.flush()
            logging.info("eval step %d %s", step, " ".join("%s %.5f" % kv for kv in scalars.items()))
    model.train(train_input_fn, max_steps=a.max_steps, writer=writer, on_checkpoint=on_checkpoint)
    dp.shutdown()


if __name__ == "__main__":
    main()
