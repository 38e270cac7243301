#!/usr/bin/env python
"""TensorFlow-checkpoint utilities for warm start / weight exchange (no TensorFlow needed; models/warm_start.py).
  list     <ckpt prefix>                         names, dtypes and shapes of a checkpoint
  template --hparam-json-file cfg.json [--hparams a=b] > map.json
                                                 a variable map of this build's parameters with placeholder TF names
  suggest  <ckpt prefix> --hparam-json-file cfg.json > map.json
                                                 the template with every placeholder replaced whose target shape occurs exactly
                                                 once in the checkpoint and once in the model (optimizer slots ignored)
  export   <model-N.pt> <out prefix> --var-map map.json --hparam-json-file cfg.json
                                                 this build's checkpoint as a TF checkpoint under the mapped names (CPU only)"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import satt_amd  # noqa: E402,F401
from satt_amd.utils.tf_checkpoint import CheckpointReader, write_checkpoint  # noqa: E402


def model_cfg(a):
    from satt_amd.hparams import hparams
    from satt_amd.params import ModelConfig
    if a.hparam_json_file:
        hparams.parse_json(open(a.hparam_json_file).read())
    hparams.parse(a.hparams)
    return ModelConfig.from_hparams(hparams)


def target_shape(cfg, tgt, shapes):
    if "buffer" in tgt:
        return None
    shp = list(shapes[tgt["param"]])
    if "rows" in tgt:
        shp[0] = tgt["rows"][1] - tgt["rows"][0]
    if "cols" in tgt:
        shp[-1] = tgt["cols"][1] - tgt["cols"][0]
    return tuple(shp)


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("cmd", choices=["list", "template", "suggest", "export"])
    ap.add_argument("args", nargs="*")
    ap.add_argument("--hparams", default="")
    ap.add_argument("--hparam-json-file", default=None)
    ap.add_argument("--var-map", default=None)
    a = ap.parse_args()
    if a.cmd == "list":
        r = CheckpointReader(a.args[0])
        for n in sorted(r.entries):
            e = r.entries[n]
            print("%-90s dtype %-2d %s" % (n, e["dtype"], e["shape"]))
        return
    from satt_amd.models.warm_start import load_var_map, template
    from satt_amd.params import param_shapes
    cfg = model_cfg(a)
    if a.cmd == "template":
        print(json.dumps(template(cfg), indent=1))
        return
    shapes = dict(param_shapes(cfg))
    if a.cmd == "suggest":
        r = CheckpointReader(a.args[0])
        slots = ("/Adam", "/Adam_1", "beta1_power", "beta2_power")
        ck = {n: tuple(e["shape"]) for n, e in r.entries.items() if not n.endswith(slots) and n != "global_step"}
        tm = template(cfg)
        by_shape_ck, by_shape_m = {}, {}
        for n, s in ck.items():
            by_shape_ck.setdefault(s, []).append(n)
        for k, t in tm.items():
            if isinstance(t, dict) and "param" in t:
                by_shape_m.setdefault(target_shape(cfg, t, shapes), []).append(k)
        out = {}
        for k, t in tm.items():
            if isinstance(t, dict) and "param" in t:
                s = target_shape(cfg, t, shapes)
                if len(by_shape_m[s]) == 1 and len(by_shape_ck.get(s, [])) == 1:
                    out[by_shape_ck[s][0]] = t
                    continue
            out[k] = t
        print(json.dumps(out, indent=1))
        return
    if a.cmd == "export":
        import torch
        st = torch.load(a.args[0], map_location="cpu")
        from satt_amd.params import layout
        lay, _ = layout(cfg)
        vm = load_var_map(a.var_map)
        flat = st["params"].numpy()
        out = {"global_step": np.array(int(st.get("step", 0)), dtype=np.int64)}
        for n, t in vm.items():
            if t.get("ignore"):
                continue
            if "buffer" in t:
                out[n] = st["bn"][t["buffer"]][0 if t["stat"] == "mean" else 1].numpy()
                continue
            o, shp = lay[t["param"]]
            v = flat[o:o + int(np.prod(shp))].reshape(shp)
            if "rows" in t:
                v = v[t["rows"][0]:t["rows"][1]]
            if "cols" in t:
                v = v[..., t["cols"][0]:t["cols"][1]]
            out[n] = np.ascontiguousarray(v)
        write_checkpoint(a.args[1], out)
        print("wrote %d variables to %s.index / .data-00000-of-00001" % (len(out), a.args[1]))


if __name__ == "__main__":
    main()
