#!/bin/bash
cd $GRAFT_REPO_ROOT
python - <<'PY'
import os, sys, numpy as np
sys.path.insert(0, '.')
import satt_amd
from satt_amd.utils import tfrecord
g = np.random.default_rng(0)
os.makedirs('/tmp/c/data', exist_ok=True); os.makedirs('/tmp/c/lists', exist_ok=True)
keys = ["LJ%03d" % i for i in range(6)]
for i, k in enumerate(keys):
    L, T = int(g.integers(8, 16)), int(g.integers(20, 40))
    s = np.concatenate([[0], g.integers(1, 60, L - 2), [0]]).astype("<i8")
    tfrecord.write_records('/tmp/c/data/%s.source.tfrecord' % k, [tfrecord.make_example({"id": i, "key": k.encode(), "source": s.tobytes(), "source_length": L, "text": b"abc"})])
    mel = g.normal(-40, 10, (T, 80)).astype("<f4")
    tfrecord.write_records('/tmp/c/data/%s.target.tfrecord' % k, [tfrecord.make_example({"id": i, "key": k.encode(), "mel": mel.tobytes(), "mel_width": 80, "target_length": T})])
open('/tmp/c/lists/train.csv','w').write("\n".join(keys[:4]) + "\n")
open('/tmp/c/lists/test.csv','w').write("\n".join(keys[4:]) + "\n")
PY
timeout 100 python train.py --max-steps 4 --source-data-root /tmp/c/data --target-data-root /tmp/c/data --checkpoint-dir /tmp/c/ck --selected-list-dir /tmp/c/lists --hparam-json-file scratch/hp_test.json --hparams "batch_size=2,save_checkpoints_steps=2,logfile=/tmp/c/log.txt,max_iters=12" 2>&1 | tail -15
echo "train rc=$?"
