"""BASELINE.json config 5 at ITS OWN workload, judged by frozen float64 vectors (tests/golden/decode_ljspeech_{b1,b8,b2}.npz, made
by tests/golden/make_decode_golden.py from oracle/torch_ref.py:infer): production dimensions, B=1 / Ti=100 (the source
bench.py:decode_bench times) and B=8 with ragged lengths, 200 FREE-RUNNING decoder steps - the feedback chain the bench runs -
through the hipGraph replay path (inference.DecodeSession, 8 steps per graph) in bf16 (the benchmark precision) and f32.
The inference branch restated: reference modules/module.py:762-778, modules/rnn_wrappers.py:47-124,188-214.

The bars below are ~3x what the MI355X measured (profiles/r05_decode_golden.log); the fixture also stores what the float64
oracle itself does when every weight matrix is rounded to bf16 (`bf16w_*`): the bf16 path is expected to sit at that size."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# (mel abs, stop abs, alignment rows abs, per-step mean |mel| abs, argmax path agreement)
# measured (MI355X, r5): bf16 mel 7.4e-4 (the float64 oracle with bf16-rounded weights: 7.7e-4 - the path sits AT the rounding floor
# of its weights), stop 1.5e-4, alignment rows 4.0e-4, drift 2.5e-5, path 0.995; f32 mode (recurrent weights are consumed as bf16 in
# both modes): mel 2.0e-4, stop 2.3e-5, alignment rows 8.0e-5, drift 1.1e-5, path 1.000
BARS = {"bf16": dict(mel=2.2e-3, stop=4.5e-4, align=1.2e-3, drift=8e-5, path=0.985),
        "f32": dict(mel=6e-4, stop=7e-5, align=2.5e-4, drift=3.5e-5, path=0.995)}
# *_sharp (r6): alignment 1 near one-hot through the 200-step feedback chain (make_decode_golden.py); a row moves by a whole position
# where two neighbouring energies are close, so the alignment / path distances are judged against the oracle's OWN bf16-weight floor
# stored in the fixture; bars ~3x measured (profiles/r06_decode_golden.log)
# measured (r6): bf16 mel 7.3e-4 / 7.4e-4 (b1_sharp / b2_sharp; the bf16-weight oracle: 7.5e-4), stop 1.3e-4, alignment rows 5.9e-3 / 1.6e-2 (a
# near one-hot row: the mass of a neighbour moves), drift 2.6e-5, path 1.000; f32 mode 2.0e-4, 1.2e-5, 1.4e-3, 5.9e-6, 1.000
BARS_SHARP = {"bf16": dict(mel=2.2e-3, stop=4.5e-4, align=5e-2, drift=8e-5, path=0.985),
              "f32": dict(mel=6e-4, stop=7e-5, align=4.5e-3, drift=3.5e-5, path=0.995)}


def _engine(z, prec, stop_shift=0.0, case=""):
    from satt_amd import ops
    from satt_amd.engine import Engine
    from satt_amd.params import ModelConfig, init_params
    cfg = ModelConfig()
    P = dict(init_params(cfg, int(z["param_seed"])))
    if case.endswith("_sharp"):
        from golden.make_bench_golden import sharpen_params
        from golden.make_decode_golden import CASES
        P = sharpen_params(P, **CASES[case]["sharpen"])
    if stop_shift:
        b = np.array(P["dec.out.b"], dtype=np.float32).copy()
        b[-1] += np.float32(stop_shift)
        P["dec.out.b"] = b
    ops.set_precision(prec)
    eng = Engine(cfg, "cuda", params=P, rng_seed=7)
    for name, (mean, var) in eng.bn.items():
        mean.copy_(torch.as_tensor(z["bn_mean." + name])); var.copy_(torch.as_tensor(z["bn_var." + name]))
    return cfg, eng


@pytest.mark.parametrize("path", ["persistent", "graph"])
@pytest.mark.parametrize("prec", ["bf16", "f32"])
@pytest.mark.parametrize("case", ["b1", "b8", "b2", "b1_sharp", "b2_sharp"])
def test_graph_decode_vs_frozen_float64_oracle(case, prec, path):
    """path: the persistent step kernel (csrc/decode_mega2.hip: bf16, B <= 2 - the benchmark's path) or the hipGraph of
    launch-per-layer steps (every other configuration, and the reference point of the persistent kernel)"""
    from satt_amd import ops
    from satt_amd.inference import infer, DecodeSession
    z = np.load(os.path.join(GOLD, "decode_ljspeech_%s.npz" % case))
    steps = int(z["steps"])
    try:
        DecodeSession.MEGA = path == "persistent"
        cfg, eng = _engine(z, prec, case=case)
        out = infer(eng, z["source"], z["source_length"], max_steps=steps, min_steps=10 ** 6, use_graph=True)
        torch.cuda.synchronize()
        took = eng._decode_sessions[next(reversed(eng._decode_sessions))].mega is not None
        if path == "persistent" and not took:
            pytest.skip("the persistent kernel does not take this case (precision / batch size)")
        assert took == (path == "persistent")
    finally:
        ops.set_precision("bf16")
        DecodeSession.MEGA = True
    assert out["steps"] == steps
    B = z["source"].shape[0]
    mel = out["mel"].float().cpu().numpy().astype(np.float64)
    stop = out["stop"].float().cpu().numpy()[..., 0].astype(np.float64)
    al1 = out["alignment1"].float().cpu().numpy(); al2 = out["alignment2"].float().cpu().numpy()
    assert np.isfinite(mel).all() and np.allclose(al1.sum(-1), 1.0, atol=1e-4) and np.allclose(al2.sum(-1), 1.0, atol=1e-4)
    sm = mel.reshape(B, steps, -1)
    rb, rt = z["rows_b"], z["rows_t"]
    e = dict(mel=np.abs(sm[rb, rt] - z["mel_rows"]).max(), stop=np.abs(stop - z["stop"]).max(),
             align=max(np.abs(al1[rb, rt] - z["align1_rows"]).max(), np.abs(al2[rb, rt] - z["align2_rows"]).max()),
             drift=np.abs(np.abs(sm).mean(-1) - z["step_abs_mel"]).max(),
             path=min((al1.argmax(-1) == z["path1"]).mean(), (al2.argmax(-1) == z["path2"]).mean()))
    if "mel" in z.files:                       # b1: every frame of the 200-step chain
        e["mel"] = max(e["mel"], np.abs(mel - z["mel"]).max())
        late = np.abs(mel - z["mel"]).reshape(steps, -1).max(-1)
        print("mel abs err by step: 0..9 %.2e, 90..99 %.2e, 190..199 %.2e" % (late[:10].max(), late[90:100].max(), late[190:].max()))
    print("[%s] " % path, end="")
    print("decode %s %s vs frozen float64 (|mel| max %.3f): mel %.3e, stop %.3e, alignment rows %.3e, per-step mean|mel| %.3e, "
          "argmax path agreement %.4f   [float64 oracle with bf16-rounded weights: mel %.2e, stop %.2e, path %.4f]"
          % (case, prec, float(z["mel_abs_max"]), e["mel"], e["stop"], e["align"], e["drift"], e["path"],
             float(z["bf16w_mel_abs_err"].max()), float(z["bf16w_stop_abs_err"].max()), float(z["bf16w_path1_agree"])))
    bar = (BARS_SHARP if case.endswith("_sharp") else BARS)[prec]
    if case.endswith("_sharp"):
        print("   [%s: alignment-1 mean row entropy %.3f nats, rows with max > 0.95: %.3f]" % (case, float(z["align1_mean_entropy"]), float(z["align1_frac_max_above_095"])))
    for k in ("mel", "stop", "align", "drift"):
        assert e[k] <= bar[k], (k, e[k], bar[k])
    assert e["path"] >= bar["path"], e["path"]


@pytest.mark.parametrize("use_graph", [True, False])
@pytest.mark.parametrize("case", ["b1", "b8", "b2"])
def test_stop_rule_fires_where_the_oracle_fires(case, use_graph):
    """the stop logit does not feed back: shifting dec.out.b[-1] by the fixture's `stop_shift` makes the rule (every sample's
    sigmoid(stop) > 0.5 and t > min_steps = 10) fire at `stop_steps` with every decision `stop_margin` clear in float64"""
    from satt_amd.inference import infer
    z = np.load(os.path.join(GOLD, "decode_ljspeech_%s.npz" % case))
    cfg, eng = _engine(z, "bf16", float(z["stop_shift"]))
    out = infer(eng, z["source"], z["source_length"], max_steps=int(z["steps"]), min_steps=10, use_graph=use_graph)
    n = int(z["stop_steps"])
    stop = out["stop"].float().cpu().numpy()[..., 0]
    err = np.abs(stop[:, :n] - (z["stop"][:, :n] + float(z["stop_shift"]))).max()
    print("stop rule %s graph=%s: engine stops after %d steps (oracle %d), stop-logit error %.2e against a margin of %.2e"
          % (case, use_graph, out["steps"], n, err, float(z["stop_margin"])))
    assert err < 0.5 * float(z["stop_margin"])
    assert out["steps"] == n and out["mel"].shape[1] == n * cfg.r
