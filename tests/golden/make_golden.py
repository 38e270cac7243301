"""Generates tests/golden/*.npz from the NumPy-float64 oracle (oracle/numpy_ref.py).

The reference (TF1 + tacotron2@6af04c7) cannot be executed here and ships no golden vectors, so
these fixtures pin the build's OWN restatement of SURVEY.md Appendix A: seeded inputs, parameters and the expected
forward outputs (dropout / zoneout ON with the counter-based masks of oracle/rng.py).  Run from the repo root:
    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from common import SMALL, MEDIUM, make_params, small_batch, oracle_cfg  # noqa: E402
from oracle import numpy_ref  # noqa: E402

CASES = {"small": (SMALL, 3, 9, 12, 7), "medium": (MEDIUM, 2, 21, 26, 11)}


def main():
    for name, (cfg_kw, B, Ti, Tm, seed) in CASES.items():
        cfg, P = make_params(cfg_kw, seed=1)
        batch = small_batch(cfg, B, Ti, Tm, seed=3)
        out = numpy_ref.forward(P, batch, oracle_cfg(cfg_kw), True, seed=seed)
        keep = {k: np.asarray(out[k], dtype=np.float32) for k in
                ("mel", "stop", "alignment1", "alignment2", "lstm_out", "sa_out", "dec_out")}
        keep.update({k: np.float64(out[k]) for k in ("loss", "mel_loss", "done_loss")})
        np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), name + ".npz"), seed=seed,
                            **{"param." + k: v for k, v in P.items()}, **{"batch." + k: v for k, v in batch.items()},
                            **{"out." + k: v for k, v in keep.items()})
        print(name, "loss", out["loss"])


if __name__ == "__main__":
    main()
