#!/usr/bin/env python
"""TensorFlow-checkpoint utilities for warm start / weight exchange (no TensorFlow needed; models/warm_start.py).
  list     <ckpt prefix>                         names, dtypes and shapes of a checkpoint
  template --hparam-json-file cfg.json [--hparams a=b] > map.json
                                                 a variable map of this build's parameters with placeholder TF names
  suggest  <ckpt prefix> --hparam-json-file cfg.json > map.json
                                                 the DEFAULT correspondence resolved against this checkpoint (what warm start
                                                 uses without a map): variable names fixed by the reference's in-tree source,
                                                 matched as suffixes, then shape uniqueness; open targets keep "?/..." keys
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
        # the default correspondence of models/warm_start.py resolved against THIS checkpoint: in-tree-derivable names by suffix
        # pattern, the rest by shape uniqueness; what stays open keeps a "?/..." placeholder for the user to fill in
        from satt_amd.models.warm_start import ShapeEngine, resolve_default_map
        r = CheckpointReader(a.args[0])
        vmap, unresolved = resolve_default_map(ShapeEngine(cfg), r)
        out = dict(vmap)
        for t in unresolved:
            tag = t.get("param") or ("%s/moving_%s" % (t["buffer"], "mean" if t["stat"] == "mean" else "variance"))
            sl = "".join("[%s %d:%d]" % (ax, t[ax][0], t[ax][1]) for ax in ("rows", "cols") if ax in t)
            out["?/%s%s" % (tag, sl)] = t
        out["global_step"] = {"ignore": True}
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
