#!/usr/bin/env python3
"""Generates the committed fixtures under tests/golden/ (run in the build container, where /root/reference exists).

Inputs  : the reference's shipped data volumes (data/*.raw, uint8), stored compressed -- data, not source.
Expected: outputs of the CPU oracle (oracle/f3d_oracle.c) on those inputs.  The reference has no tests or golden
          vectors of its own, so these pin the GPU path to the oracle at sizes the oracle cannot
          be re-run at inside a quick test, and detect drift of either side.

  python tests/golden/make_golden.py            # ~5 minutes on 8 cores
  python tests/golden/make_golden.py --piecemeal   # only expected_piecemeal.npz (seconds): the two crops through the pipeline
                                                   # of the reference's piecemeal driver (no pre-blur, no median)
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402

REF_DATA = "/root/reference/data"


def load_u8(name, shape):
    return np.fromfile(os.path.join(REF_DATA, name), np.uint8).reshape(shape)


def summary(vol):
    v = vol.astype(np.float64)
    return np.array([v.sum(), np.sqrt((v * v).sum()), v.min(), v.max()])


def digest(*vols):
    h = hashlib.sha256()
    for v in vols:
        h.update(np.ascontiguousarray(v + np.float32(0.0)).tobytes())  # +0.0 canonicalises the sign of zero
    return h.hexdigest()


def piecemeal_only():
    """The out-of-core driver's semantics (src/optical_flow/optical_flow_p.cpp: no Gaussian, median commented out) on the
    two crops of the whole-pipeline fixtures; inputs come from the committed inputs_*.npz."""
    i128 = np.load(os.path.join(HERE, "inputs_128.npz"))
    irub = np.load(os.path.join(HERE, "inputs_rub.npz"))
    F0, F1 = i128["frame_0"].astype(np.float32), i128["frame_1"].astype(np.float32)
    R0 = np.repeat(irub["slice_0"][None], int(irub["depth"]), axis=0).astype(np.float32)
    R1 = np.repeat(irub["slice_1"][None], int(irub["depth"]), axis=0).astype(np.float32)
    out = {}
    crop = (slice(40, 64), slice(40, 80), slice(40, 88))
    (u, v, w), lv = orc.compute_flow(F0[crop].copy(), F1[crop].copy(), gaussian_sigma=0.0, median_radius=1)
    out.update(crop128_flow=np.stack([u, v, w]), crop128_levels=lv)
    rc = (slice(0, 5), slice(100, 164), slice(200, 296))
    (u, v, w), lv = orc.compute_flow(R0[rc].copy(), R1[rc].copy(), gaussian_sigma=0.0, median_radius=1)
    out.update(croprub_flow=np.stack([u, v, w]), croprub_levels=lv)
    np.savez_compressed(os.path.join(HERE, "expected_piecemeal.npz"), **out)
    print("piecemeal crops done", out["crop128_levels"], out["croprub_levels"])


def main():
    if "--piecemeal" in sys.argv:
        return piecemeal_only()
    f0 = load_u8("frame_0_128-128-128.raw", (128, 128, 128))
    f1 = load_u8("frame_1_128-128-128.raw", (128, 128, 128))
    r0 = load_u8("rub1-584-388-5.raw", (5, 388, 584))
    r1 = load_u8("rub2-584-388-5.raw", (5, 388, 584))
    assert all(np.array_equal(r0[0], r0[k]) and np.array_equal(r1[0], r1[k]) for k in range(5))
    np.savez_compressed(os.path.join(HERE, "inputs_128.npz"), frame_0=f0, frame_1=f1)
    np.savez_compressed(os.path.join(HERE, "inputs_rub.npz"), slice_0=r0[0], slice_1=r1[0], depth=5)

    F0, F1 = f0.astype(np.float32), f1.astype(np.float32)
    R0, R1 = r0.astype(np.float32), r1.astype(np.float32)
    out = {}

    # C1: 128^3, one level, 1 outer x 5 inner, other parameters default
    (u, v, w), _ = orc.compute_flow(F0, F1, warp_levels_count=1, outer_iterations_count=1, inner_iterations_count=5)
    c = slice(48, 80)
    out.update(c1_crop=np.stack([u[c, c, c], v[c, c, c], w[c, c, c]]), c1_summary=np.stack([summary(a) for a in (u, v, w)]))
    out["c1_sha256"] = digest(u, v, w)
    print("C1 done")

    # whole-pipeline crops, full defaults
    crop = (slice(40, 64), slice(40, 80), slice(40, 88))  # 48 x 40 x 24
    (u, v, w), lv = orc.compute_flow(F0[crop].copy(), F1[crop].copy())
    out.update(crop128_flow=np.stack([u, v, w]), crop128_levels=lv)
    rc = (slice(0, 5), slice(100, 164), slice(200, 296))  # 96 x 64 x 5
    (u, v, w), lv = orc.compute_flow(R0[rc].copy(), R1[rc].copy())
    out.update(croprub_flow=np.stack([u, v, w]), croprub_levels=lv)
    print("crops done")

    # C2 / C3 full size, full defaults: checksums, norms and two orthogonal centre slices per component
    for tag, (A, B) in (("c2", (F0, F1)), ("c3", (R0, R1))):
        (u, v, w), lv = orc.compute_flow(A, B)
        d, h, _ = u.shape
        out[tag + "_levels"] = lv
        out[tag + "_sha256"] = digest(u, v, w)
        out[tag + "_summary"] = np.stack([summary(a) for a in (u, v, w)])
        zs = np.stack([a[d // 2] for a in (u, v, w)])
        out[tag + "_slice_z"] = zs if tag == "c2" else zs[:, 130:258, 228:356]  # C3: a 128 x 128 centre window
        out[tag + "_slice_y"] = np.stack([a[:, h // 2] for a in (u, v, w)])
        print(tag, "done", lv, [float(np.abs(a).max()) for a in (u, v, w)])
    np.savez_compressed(os.path.join(HERE, "expected_oracle.npz"), **out)


if __name__ == "__main__":
    main()
