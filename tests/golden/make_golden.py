"""Regenerates tests/golden/*.npz from the CPU oracle (run from the repo root:
`python tests/golden/make_golden.py`).  The reference itself cannot run in this image (CUDA), so the
fixtures are oracle outputs on seeded inputs; the oracle is pinned to the reference in
tests/test_oracle_pins.py."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "gpu-computing-course_amd", "pyhost")]
import mi355_synth as synth  # noqa: E402
import oracle  # noqa: E402


def main():
    out = {}
    for name, (verts, vidx) in (("soup_5k", synth.soup(5000, 0.06, 42)), ("cloth_30", synth.cloth_pair(30))):
        r = oracle.pipeline(verts, vidx)
        out[name + "_pairs"] = oracle.pair_set(r["pairs"])
        out[name + "_tested"] = np.uint64(r["stats"].pairs_tested)
        out[name + "_keys_head"] = r["keys"][:64]
        print(name, len(out[name + "_pairs"]), "pairs,", int(out[name + "_tested"]), "tested")
    np.savez_compressed(os.path.join(HERE, "cd_golden.npz"), **out)
    spheres, shifts = synth.sphere_scene(96, 256, seed=11)
    shifts[:, 0] = (np.arange(96) * 7) % 23 - 11
    shifts[:, 1] = (np.arange(96) * 5) % 19 - 9
    img = oracle.rt_render(spheres, shifts, 256, 3, -2)
    np.savez_compressed(os.path.join(HERE, "rt_golden.npz"), img=img)
    print("rt frame checksum", int(img.astype(np.uint64).sum()))


if __name__ == "__main__":
    main()
