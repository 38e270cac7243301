"""The configuration surface pinned to the reference: tests/golden/reference_hparams.json is DATA extracted from
/root/reference by tools/make_reference_fixtures.py (hparams.py:10-226 parsed with ast, examples/*/*.json loaded) - the one
part of the reference that can be read without TensorFlow.  Every key, default and type of the reference's schema and every
model-selection key of its four example configurations must be reproduced by this build."""
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_hparams.json")))
EXTENSIONS = {"warm_start_var_map"}      # documented additions of this build (hparams.py)


def test_every_reference_hparam_with_its_default_and_type():
    from satt_amd.hparams import hparams
    ours = hparams.values()
    ref = FIX["defaults"]
    assert len(ref) == 156                                   # hparams.py:10-226
    assert set(ref) - set(ours) == set(), "reference keys missing here"
    assert set(ours) - set(ref) == EXTENSIONS
    for k, v in ref.items():
        o = ours[k]
        o = list(o) if isinstance(o, tuple) else o
        assert o == v, (k, o, v)
        assert type(o).__name__ == FIX["types"][k], (k, type(o).__name__, FIX["types"][k])


@pytest.mark.parametrize("name", sorted(FIX["examples"]))
def test_example_configs_carry_the_reference_selection_keys(name):
    """examples/<corpus>/<model>.json: every scalar key of the reference's file with the same value (the per-bin mel statistics
    are corpus data and are not shipped: they come from the preprocessing run's hparams.json)"""
    ref = FIX["examples"][name]
    ours = json.load(open(os.path.join(ROOT, "examples", name)))
    ours = {k: v for k, v in ours.items() if not k.startswith("_")}
    assert set(ref["lists"]) == {"average_mel_level_db", "stddev_mel_level_db"}
    assert set(ours) == set(ref["scalars"]), (set(ours) ^ set(ref["scalars"]))
    for k, v in ref["scalars"].items():
        assert ours[k] == v and type(ours[k]) in (type(v), float, int), (k, ours[k], v)


@pytest.mark.parametrize("name", sorted(FIX["examples"]))
def test_example_configs_resolve_like_the_reference(name):
    """defaults <- example JSON gives the same effective values as the reference's defaults <- the reference's JSON, for
    every key of the schema except the two corpus tables"""
    from satt_amd.hparams import hparams
    hp = hparams.copy().parse_json(open(os.path.join(ROOT, "examples", name)).read())
    want = dict(FIX["defaults"])
    want.update(FIX["examples"][name]["scalars"])
    got = hp.values()
    for k, v in want.items():
        if k in ("average_mel_level_db", "stddev_mel_level_db"):
            continue
        o = got[k]
        o = list(o) if isinstance(o, tuple) else o
        assert o == v, (k, o, v)
        assert isinstance(o, bool) == isinstance(FIX["defaults"][k], bool), k


def test_default_mel_statistics_are_refused_not_divided_by():
    """the schema's default stddev is [0.0] (reference hparams.py:21): prepare_target must refuse it instead of producing
    inf / NaN targets"""
    from satt_amd.datasets import ljspeech
    from satt_amd.hparams import hparams
    hp = hparams.copy().parse_json(open(os.path.join(ROOT, "examples", "ljspeech", "tacotron.json")).read())
    t = dict(id=1, key="k", mel=np.zeros((6, 80), np.float32), mel_width=80, target_length=6)
    with pytest.raises(ValueError, match="stddev_mel_level_db"):
        ljspeech.prepare_target(t, hp)
    hp.parse_json({"average_mel_level_db": [0.0] * 7, "stddev_mel_level_db": [1.0] * 7})
    with pytest.raises(ValueError, match="entries"):
        ljspeech.prepare_target(t, hp)
    hp.parse_json({"average_mel_level_db": [-40.0] * 80, "stddev_mel_level_db": [10.0] * 80})
    assert np.isfinite(ljspeech.prepare_target(t, hp).mel).all()
