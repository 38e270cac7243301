#!/usr/bin/env python3
"""Pins the configuration surface to the reference WITHOUT TensorFlow (which cannot be imported here): the only part of
/root/reference that can be read as data is its hyper-parameter schema and its example configurations.

Run in the BUILD container (the reference tree does not exist on the GPU box):

    python tools/make_reference_fixtures.py [--reference /root/reference]

* `hparams.py:10-226` is parsed with `ast` (never imported): the keyword arguments of the
  `tf.contrib.training.HParams(...)` call, evaluated with `ast.literal_eval` -> {name: default}.
* `examples/{ljspeech,vctk}/*.json` are loaded as JSON.  The 80-bin mel statistics tables (lists) are data of the corpora,
  not of the model: they are reduced to their length and a CRC-32 so that the fixture stays small and no table is copied;
  every scalar / string key (the model-selection keys of `examples/ljspeech/self-attention-tacotron.json:171-185`) is kept.

Output: tests/golden/reference_hparams.json (data only: names, defaults, selection keys), checked by
tests/test_reference_fixtures_cpu.py against self-attention-tacotron_amd/hparams.py and examples/.
"""
import argparse
import ast
import json
import os
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def parse_hparams(path):
    tree = ast.parse(open(path).read(), filename=path)
    for node in ast.walk(tree):
        if isinstance(node, ast.Call) and getattr(node.func, "attr", "") == "HParams":
            out = {}
            for kw in node.keywords:
                v = ast.literal_eval(kw.value)
                out[kw.arg] = list(v) if isinstance(v, tuple) else v
            return out, node.lineno, node.end_lineno
    raise SystemExit("no HParams(...) call in %s" % path)


def list_digest(v):
    """length + CRC-32 of the canonical JSON text of a list-valued key"""
    return {"len": len(v), "crc32": zlib.crc32(json.dumps(v, separators=(",", ":")).encode())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "reference_hparams.json"))
    a = ap.parse_args()
    defaults, l0, l1 = parse_hparams(os.path.join(a.reference, "hparams.py"))
    examples = {}
    for corpus in ("ljspeech", "vctk"):
        d = os.path.join(a.reference, "examples", corpus)
        for fn in sorted(os.listdir(d)):
            if not fn.endswith(".json"):
                continue
            cfg = json.load(open(os.path.join(d, fn)))
            examples["%s/%s" % (corpus, fn)] = {
                "scalars": {k: v for k, v in cfg.items() if not isinstance(v, list)},
                "lists": {k: list_digest(v) for k, v in cfg.items() if isinstance(v, list)},
            }
    fixture = {
        "_made_by": "tools/make_reference_fixtures.py",
        "_source": "hparams.py:%d-%d (ast, not imported); examples/*/*.json" % (l0, l1),
        "defaults": defaults,
        "types": {k: type(v).__name__ for k, v in defaults.items()},
        "examples": examples,
    }
    with open(a.out, "w") as f:
        json.dump(fixture, f, indent=1, sort_keys=True)
        f.write("\n")
    print("wrote %s: %d hparams, %d example configs" % (a.out, len(defaults), len(examples)))


if __name__ == "__main__":
    main()
