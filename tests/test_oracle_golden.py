"""The committed golden images (tests/golden/, made by make_golden.py with the
oracle's faithful BVH mode) are reproduced by the oracle on row bands — guards
the fixtures against drift of the oracle, the models or the sample generator.
CPU only; the GPU parity tests compare the HIP path with the same files."""
import json
import os

import numpy as np
import pytest
from PIL import Image

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return np.asarray(Image.open(os.path.join(GOLD, name + ".png")).convert("RGB"))


def golden_meta():
    with open(os.path.join(GOLD, "golden.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("name,bands", [
    ("c1_bunny_256_seed", [(0, 256)]),
    ("c1b_bigbunny_256_seed", [(60, 6), (150, 6), (236, 6)]),
    ("c1b_bigbunny_256_half", [(100, 6), (200, 6)]),
    ("ragged_bigbunny_203x117_seed", [(50, 5), (112, 5)]),
])
def test_oracle_reproduces_golden_bands(orc, samples_seeded, samples_half, name, bands):
    meta = golden_meta()
    case = meta["cases"][name]
    assert meta["seed"] == orc.SEED
    gold = load_golden(name)
    assert gold.shape == (case["height"], case["width"], 3)
    assert int(gold.astype(np.uint64).sum()) == case["byte_sum"]
    T = samples_seeded if case["table"] == "seed" else samples_half
    s = orc.default_scene(case["objs"], case["width"], case["height"], T)
    for row0, nrows in bands:
        img, st = s.render_rows(row0, nrows, mode=orc.MODE_BVH)
        assert np.array_equal(img, gold[row0:row0 + nrows]), (name, row0)
        if nrows == case["height"]:
            assert st["primary_hits"] == case["primary_hits"]
            assert st["primary_rays"] + st["shadow_rays"] == case["r_total"]
