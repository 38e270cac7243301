"""CPU checks of tests/golden/bench_*.npz (the float64 oracle frozen at the benchmark workloads by
tests/golden/make_bench_golden.py): the fixtures belong to the inputs the GPU test rebuilds (seeded batch + init_params),
are self-consistent, and the count sketch they store the gradient in measures distances the way the GPU test assumes."""
import os

import numpy as np
import pytest

from common import count_sketch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", ["ljspeech", "vctk", "ljspeech_sharp_lo", "ljspeech_sharp_hi"])
def test_fixture_belongs_to_the_seeded_inputs_and_is_self_consistent(name, satt):
    from golden.make_bench_golden import CASES, crc_of, make_batch, sample_rows, sharpen_params
    from satt_amd.params import ModelConfig, init_params, param_shapes
    z = np.load(os.path.join(GOLD, "bench_%s.npz" % name))
    case = CASES[name]
    cfg = ModelConfig(**case["cfg"])
    batch = make_batch(case["batch"])
    assert crc_of(batch) == int(z["meta.batch_crc"])
    P = init_params(cfg, int(z["meta.param_seed"]))
    if "sharpen" in case:
        # the converged-regime fixtures (r6): the SAME seeded model with its location-sensitive score reshaped; the regime they were
        # frozen in is part of the fixture - near one-hot rows, on either side of the kernels' softmax-form switch at sum|v| = 30
        P = sharpen_params(P, **case["sharpen"])
        assert float(z["align1_mean_entropy"]) < 0.7 and float(z["align1_max_mean"]) > 0.75
        sv = float(np.abs(np.asarray(P["dec.att1.v"], dtype=np.float64)).sum())
        assert abs(sv - float(z["att1_v_abs_sum"])) < 1e-3 * sv and ((sv < 30.0) if name.endswith("_lo") else (sv > 100.0))
        if name.endswith("_hi"):
            assert float(z["align1_frac_max_above_095"]) > 0.7          # max alpha > 0.95 on most rows
        base = np.load(os.path.join(GOLD, "bench_ljspeech.npz"))
        assert float(base["align1_mean_entropy"]) > 2.5                # (the diffuse regime every earlier fixture sits in)
    assert crc_of(P) == int(z["meta.param_crc"])
    B, Td = batch["done"].shape
    Ti = batch["source"].shape[1]
    assert (B, Ti, Td * cfg.r) == (case["batch"]["B"], case["batch"]["Ti"], case["batch"]["Tm"])
    assert abs(float(z["loss"]) - float(z["mel_loss"]) - float(z["done_loss"])) < 1e-12
    # the batch-level mel-L1 is the mask-weighted mean of the per-sample ones
    n = batch["spec_loss_mask"].sum(-1)
    assert abs(float((z["per_sample_mel_l1"] * n).sum() / n.sum()) - float(z["mel_loss"])) < 1e-9
    sb, st = sample_rows(B, Td, 99)
    assert np.array_equal(sb, z["rows_b"]) and np.array_equal(st, z["rows_t"])
    for k in ("align1_rows", "align2_rows"):
        a = z[k]
        assert a.shape == (int(z["meta.nrow"]), Ti) and (a >= 0).all() and np.allclose(a.sum(-1), 1.0, atol=1e-5)
        for i, b in enumerate(sb):                         # no mass beyond the sample's memory length
            assert np.all(a[i, int(batch["source_length"][b]):] == 0)
    assert np.array_equal(z["align1_rows"].argmax(-1), z["path1"][sb, st])
    assert z["path1"].shape == (B, Td)
    if "sharpen" not in case:
        assert z["path1"][:, 0].max() <= 1          # alpha_0 = onehot(0): step 0 cannot leave rows 0..1
    else:       # ... unless an energy gap > log(0.5 / 1e-7) = 15.4 lets the recursion's `+ 1e-7` floor win: the regime these fixtures pin
        assert z["path1"][:, 0].max() > 1
    assert z["mel_rows"].shape == (int(z["meta.nrow"]), cfg.r * cfg.num_mels)
    names = [str(s) for s in z["grad_names"]]
    assert names == [k for k, _ in param_shapes(cfg)]
    assert abs(np.sqrt((z["grad_norms"] ** 2).sum()) - float(z["grad_norm_all"])) < 1e-9
    shapes = dict(param_shapes(cfg))
    for i, k in enumerate(names):
        size = int(np.prod(shapes[k]))
        if size <= int(z["meta.full_max"]):
            g = z["grad_full." + k]
            assert g.shape == tuple(shapes[k]) and abs(np.linalg.norm(g) - z["grad_norms"][i]) <= 1e-5 * z["grad_norms"][i] + 1e-12
        else:
            sk = z["grad_sketch." + k]
            assert sk.shape == (int(z["meta.sketch_t"]),)
            # a count sketch preserves the norm in expectation (variance ~ 2/dim of the squared norm)
            assert abs(np.linalg.norm(sk) / max(z["grad_norms"][i], 1e-30) - 1.0) < 0.2, k
    assert abs(np.linalg.norm(z["grad_sketch_all"]) / float(z["grad_norm_all"]) - 1.0) < 0.05
    if name == "vctk":                                     # only the batch's speakers receive an embedding gradient
        assert "grad_sketch.speaker_embedding" in z.files or "grad_full.speaker_embedding" in z.files


def test_count_sketch_measures_relative_distance():
    g = np.random.default_rng(0)
    a = g.normal(size=300_000)
    for eps in (1e-3, 3e-2):
        b = a + eps * g.normal(size=a.size)
        true = np.linalg.norm(a - b) / np.linalg.norm(a)
        est = np.linalg.norm(count_sketch(a, 4096, 0) - count_sketch(b, 4096, 0)) / np.linalg.norm(count_sketch(a, 4096, 0))
        assert abs(est / true - 1.0) < 0.1, (eps, true, est)
    assert not np.allclose(count_sketch(a, 1024, 3), count_sketch(a, 1024, 4))       # the salt selects the projection
