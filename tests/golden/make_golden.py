#!/usr/bin/env python3
"""Generate the committed golden images with the CPU oracle (faithful BVH mode).

The reference cannot be built or run (Rust; no toolchain) and ships no golden
for its current code, so these vectors come from the oracle, which is itself
pinned only by tests/test_oracle_kat.py and tests/test_oracle_numpy.py
("parity unpinned", see oracle/oracle.h).  Run from the repo root:

    python tests/golden/make_golden.py

Outputs (tests/golden/): PNGs of the decoded RGB8 framebuffer + golden.json
with the ray counts the GPU path must reproduce.
"""
import json
import os
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import orclib as o  # noqa: E402

CASES = [
    # name, objs, W, H, table
    ("c1_bunny_256_seed", ["bunny.obj"], 256, 256, "seed"),
    ("c1b_bigbunny_256_seed", ["big_bunny.obj"], 256, 256, "seed"),
    ("c1b_bigbunny_256_half", ["big_bunny.obj"], 256, 256, "half"),
    ("ragged_bigbunny_203x117_seed", ["big_bunny.obj"], 203, 117, "seed"),
]


def main():
    tables = {"seed": o.gen_samples(), "half": o.const_samples(0.5)}
    meta = {"seed": o.SEED, "n_samples": o.NB_RAND_SAMPLE, "cases": {}}
    for name, objs, W, H, tab in CASES:
        s = o.default_scene(objs, W, H, tables[tab])
        img, st, tri = s.render_rows(mode=o.MODE_BVH, want_tri=True)
        Image.fromarray(img, "RGB").save(os.path.join(HERE, name + ".png"), optimize=True)
        ground = int((tri == s.n_tris - 1).sum())
        meta["cases"][name] = {
            "objs": objs, "width": W, "height": H, "table": tab,
            "primary_hits": int(st["primary_hits"]), "mesh_hits": int(st["mesh_hits"]), "ground_hits": ground,
            "r_total": int(st["primary_rays"] + st["shadow_rays"]),
            "exact_ties": int(st["exact_ties"]), "nonfinite_t": int(st["nonfinite_t"]),
            "assert_tmin_gt_tmax": int(st["assert_tmin_gt_tmax"]),
            "byte_sum": int(img.astype(np.uint64).sum()),
        }
        print(name, meta["cases"][name], "%.1fs" % (st["render_ms"] / 1e3), flush=True)
        s.close()
    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
