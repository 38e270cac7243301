"""CPU checks of tests/golden/decode_ljspeech_*.npz (float64 free-running decode frozen by make_decode_golden.py): the fixtures
belong to the seeded inputs the GPU test rebuilds, are self-consistent, and a SHORT live re-run of the oracle reproduces their
first steps (the fixture is the oracle's output, not something else's)."""
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("case", ["b1", "b8", "b2", "b1_sharp", "b2_sharp"])
def test_decode_fixture_is_consistent_and_reproducible(case, satt):
    from golden.make_decode_golden import CASES, STEPS, decode_inputs, moving_stats, pick_stop_shift, stop_rule_step
    from oracle import torch_ref
    from satt_amd.params import ModelConfig, init_params
    z = np.load(os.path.join(GOLD, "decode_ljspeech_%s.npz" % case))
    cfg = ModelConfig()
    B, Ti = CASES[case]["B"], CASES[case]["Ti"]
    src, sl = decode_inputs(B, Ti)
    assert np.array_equal(src, z["source"]) and np.array_equal(sl, z["source_length"]) and int(z["steps"]) == STEPS
    if B == 1:      # the source bench.py:decode_bench draws
        g = np.random.default_rng(1234)
        s = g.integers(1, 68, (1, Ti)); s[:, 0] = 0; s[:, -1] = 0
        assert np.array_equal(s, src)
    for k, (m, v) in moving_stats(cfg).items():
        assert np.array_equal(m, z["bn_mean." + k]) and np.array_equal(v, z["bn_var." + k])
    assert z["stop"].shape == (B, STEPS) and z["path1"].shape == (B, STEPS)
    if "sharpen" not in CASES[case]:
        assert z["path1"][:, 0].max() <= 1                   # alpha_0 = onehot(0): step 0 cannot leave rows 0..1
    # (with energy gaps > log(0.5 / 1e-7) = 15.4 it CAN: the `+ 1e-7` of the recursion times a huge softmax weight beats the two
    #  rows the one-hot start reaches - b2_sharp's first sample jumps to row 15 at step 0; that floor is what the sharp cases pin)
    for b in range(B):                                       # the argmax path never leaves the sample's memory
        assert z["path1"][b].max() < sl[b] and z["path2"][b].max() < sl[b]
    for k in ("align1_rows", "align2_rows"):
        a = z[k]
        assert (a >= 0).all() and np.allclose(a.sum(-1), 1.0, atol=1e-5)
        for i, b in enumerate(z["rows_b"]):
            assert np.all(a[i, sl[b]:] == 0)
    assert np.array_equal(z["align1_rows"].argmax(-1), z["path1"][z["rows_b"], z["rows_t"]])
    shift, n, margin = pick_stop_shift(z["stop"].astype(np.float64))
    assert n == int(z["stop_steps"]) and abs(shift - float(z["stop_shift"])) < 1e-6 and margin > 10 * float(z["bf16w_stop_abs_err"].max())
    assert stop_rule_step(z["stop"].astype(np.float64), float(z["stop_shift"])) == n
    if "mel" in z.files:
        sm = z["mel"].reshape(B, STEPS, -1)
        assert np.allclose(np.abs(sm).mean(-1), z["step_abs_mel"], atol=1e-6)
        assert np.array_equal(sm[z["rows_b"], z["rows_t"]], z["mel_rows"])
    # live oracle, first 12 steps
    P = init_params(cfg, int(z["param_seed"]))
    if "sharpen" in CASES[case]:          # the converged-regime cases (r6): near one-hot alignment 1 through the whole chain
        from golden.make_bench_golden import sharpen_params
        P = sharpen_params(P, **CASES[case]["sharpen"])
        assert float(z["align1_mean_entropy"]) < 0.2 and float(z["align1_frac_max_above_095"]) > 0.95
    mv = {k: (torch.as_tensor(z["bn_mean." + k], dtype=torch.float64), torch.as_tensor(z["bn_var." + k], dtype=torch.float64))
          for k in ("bank", "proj1", "proj2")}
    ref = torch_ref.infer(torch_ref.to_torch(P), torch.as_tensor(src), torch.as_tensor(sl), torch_ref.Cfg(), 12, mv, min_steps=10 ** 6)
    assert np.abs(ref["stop"].numpy()[..., 0] - z["stop"][:, :12]).max() < 1e-6
    assert np.array_equal(ref["alignment1"].numpy().argmax(-1), z["path1"][:, :12])
