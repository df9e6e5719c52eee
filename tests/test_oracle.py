"""The oracle against the committed golden vectors and against properties the domain offers (no GPU).
Pinning status: the reference ships no known-answer tests for its kernels, so apart from the level schedule and the
volume I/O (tests/test_host_logic.py, checked against the reference's own sources) the oracle's kernel numerics are
'parity unpinned' (DESIGN.md); these tests keep it self-consistent and keep the fixtures honest."""
import os

import numpy as np
import pytest

from conftest import bit_same, box_in_container

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def gold():
    e = np.load(os.path.join(GOLD, "expected_oracle.npz"))
    i128 = np.load(os.path.join(GOLD, "inputs_128.npz"))
    irub = np.load(os.path.join(GOLD, "inputs_rub.npz"))
    return dict(e=e, f0=i128["frame_0"].astype(np.float32), f1=i128["frame_1"].astype(np.float32),
                r0=np.repeat(irub["slice_0"][None], 5, 0).astype(np.float32),
                r1=np.repeat(irub["slice_1"][None], 5, 0).astype(np.float32))


def test_fixture_inputs_are_the_shipped_volumes(gold):
    import hashlib
    # sha256 of the reference's data files, recomputed from the stored fixtures
    i128 = np.load(os.path.join(GOLD, "inputs_128.npz"))
    irub = np.load(os.path.join(GOLD, "inputs_rub.npz"))
    assert hashlib.sha256(i128["frame_0"].tobytes()).hexdigest() == "a795cd6a36609f53e4839716a606546a85a555bad9760bf9b0275559e7753220"
    assert hashlib.sha256(i128["frame_1"].tobytes()).hexdigest() == "f2f7b1fbe77ba64a3bb6a794213c2b8d3a42c575cc729b67b10e1307c39bc06b"
    rub1 = np.repeat(irub["slice_0"][None], 5, 0)
    rub2 = np.repeat(irub["slice_1"][None], 5, 0)
    assert hashlib.sha256(rub1.tobytes()).hexdigest() == "fa83c985aa137cccd46f537d949e09e80b5b0bf9511a695b5982f9b921735e8c"
    assert hashlib.sha256(rub2.tobytes()).hexdigest() == "b3e7ba3ef3b821d13eaa0d1fd2be5db1a426a631c0d5cb00a0731273b930d56d"


def test_oracle_reproduces_the_golden_crops(oracle, gold):
    crop = (slice(40, 64), slice(40, 80), slice(40, 88))
    (u, v, w), levels = oracle.compute_flow(gold["f0"][crop].copy(), gold["f1"][crop].copy())
    assert levels == int(gold["e"]["crop128_levels"])
    assert bit_same(np.stack([u, v, w]), gold["e"]["crop128_flow"])
    rc = (slice(0, 5), slice(100, 164), slice(200, 296))
    (u, v, w), levels = oracle.compute_flow(gold["r0"][rc].copy(), gold["r1"][rc].copy())
    assert bit_same(np.stack([u, v, w]), gold["e"]["croprub_flow"])


def test_oracle_reproduces_the_piecemeal_crops(oracle, gold):
    """The pipeline of the reference's out-of-core driver (no pre-blur, no median: optical_flow_p.cpp) on the same crops."""
    e = np.load(os.path.join(GOLD, "expected_piecemeal.npz"))
    crop = (slice(40, 64), slice(40, 80), slice(40, 88))
    (u, v, w), levels = oracle.compute_flow(gold["f0"][crop].copy(), gold["f1"][crop].copy(), gaussian_sigma=0.0, median_radius=1)
    assert levels == int(e["crop128_levels"])
    assert bit_same(np.stack([u, v, w]), e["crop128_flow"])
    rc = (slice(0, 5), slice(100, 164), slice(200, 296))
    (u, v, w), levels = oracle.compute_flow(gold["r0"][rc].copy(), gold["r1"][rc].copy(), gaussian_sigma=0.0, median_radius=1)
    assert bit_same(np.stack([u, v, w]), e["croprub_flow"])
    # and the two pipelines really differ: the fixture pins the piecemeal semantics, not a copy of the other one
    assert not np.array_equal(e["crop128_flow"], gold["e"]["crop128_flow"])


def test_padded_container_gives_the_same_flow(oracle, gold):
    crop = (slice(50, 62), slice(40, 64), slice(40, 72))
    a, _ = oracle.compute_flow(gold["f0"][crop].copy(), gold["f1"][crop].copy(), warp_levels_count=6, outer_iterations_count=3)
    b, _ = oracle.compute_flow(gold["f0"][crop].copy(), gold["f1"][crop].copy(), pitch_f=48, warp_levels_count=6,
                               outer_iterations_count=3)
    for x, y in zip(a, b):
        assert bit_same(x, y)  # NaN-poisoned padding never reaches the sub-box


def test_known_translation_is_recovered(oracle, f3d):
    f0, f1 = f3d.synth_pair(40, 36, 32)
    (u, v, w), _ = oracle.compute_flow(f0, f1)
    inner = (slice(8, -8),) * 3
    assert abs(float(u[inner].mean()) - 2.0) < 0.15
    assert abs(float(v[inner].mean()) + 1.0) < 0.15
    assert abs(float(w[inner].mean()) - 0.5) < 0.15


def test_identical_frames_give_zero_flow(oracle, f3d):
    f0, _ = f3d.synth_pair(32, 24, 16)
    (u, v, w), _ = oracle.compute_flow(f0, f0.copy(), warp_levels_count=5, outer_iterations_count=3)
    assert not np.abs(u).any() and not np.abs(v).any() and not np.abs(w).any()


def test_median_is_the_window_median(oracle):
    rng = np.random.default_rng(1)
    dims = (11, 9, 7)
    W, H, D = dims
    vol = rng.normal(size=(D, H, W)).astype(np.float32)
    for r in (3, 5, 7):
        h = r // 2
        pad = np.pad(vol, h, mode="reflect")  # mirror without repeating the edge, like the reference's halo loads
        win = np.lib.stride_tricks.sliding_window_view(pad, (r, r, r)).reshape(D, H, W, -1)
        expected = np.sort(win, axis=-1)[..., (r ** 3) // 2]
        assert np.array_equal(oracle.median(vol, dims, r), expected)


def test_resample_warp_and_blur_invariants(oracle):
    dims, cdims = (21, 13, 9), (32, 16, 12)
    W, H, D = dims
    const = np.full((cdims[2], cdims[1], cdims[0]), np.nan, np.float32)
    const[:D, :H, :W] = 3.25
    out = oracle.resample(const, dims, (17, 11, 8))
    assert np.allclose(out[:8, :11, :17], 3.25, rtol=2e-5)          # area resampling preserves constants
    out = oracle.resample(const, dims, (24, 15, 11))
    assert np.allclose(out[:11, :15, :24], 3.25, rtol=2e-5)
    rng = np.random.default_rng(2)
    f0 = box_in_container(rng, dims, cdims, 0, 255)
    f1 = box_in_container(rng, dims, cdims, 0, 255)
    zero = np.full_like(f0, np.nan)
    zero[:D, :H, :W] = 0
    assert bit_same(oracle.warp(f0, f1, zero, zero, zero, dims, (1.5, 1.0, 2.0))[:D, :H, :W], f1[:D, :H, :W])
    far = zero.copy()
    far[:D, :H, :W] = 1e6                                             # every target leaves the domain -> frame_0
    assert bit_same(oracle.warp(f0, f1, far, zero, zero, dims, (1.0, 1.0, 1.0))[:D, :H, :W], f0[:D, :H, :W])
    blur_dims, blur_c = (40, 16, 30), (44, 16, 30)
    c2 = np.full((blur_c[2], blur_c[1], blur_c[0]), np.nan, np.float32)
    c2[:, :, :40] = 10.0
    g = oracle.gaussian(c2, blur_dims, 2.0)
    assert np.allclose(g[6:-6, 6:-6, 6:34], 10.0, rtol=1e-5)       # interior untouched, borders darkened by zero padding
    assert float(g[0, 0, 0]) < 10.0 * 0.6 ** 3 + 1.0


def test_kernels_on_a_z_window_equal_the_whole_volume(oracle):
    """The slab parameterisation the multi-GPU driver relies on, on the oracle itself."""
    rng = np.random.default_rng(4)
    dims, cdims = (19, 11, 14), (24, 12, 14)
    W, H, D = dims
    h = (1.2, 0.9, 1.7)
    arrs = [box_in_container(rng, dims, cdims, lo, hi) for lo, hi in
            [(0, 255), (0, 255), (-2, 2), (-2, 2), (-2, 2), (-.5, .5), (-.5, .5), (-.5, .5)]]
    phi, ksi = oracle.phi_ksi(*arrs, dims, h, 0.001, 0.001)
    whole = oracle.solve_sweep(*arrs, phi, ksi, dims, h, 7.5)
    z_lo, z_hi, z_base = 4, 9, 3                        # container plane 0 holds global plane 3
    sub = lambda a: np.ascontiguousarray(a[z_base:z_hi + 1])
    g = oracle.Geom(cdims[1], cdims[0], z_base, z_lo, z_hi)
    p2, k2 = oracle.phi_ksi(*[sub(a) for a in arrs], dims, h, 0.001, 0.001, g=g)
    assert bit_same(p2[z_lo - z_base:z_hi - z_base, :H, :W], phi[z_lo:z_hi, :H, :W])
    part = oracle.solve_sweep(*[sub(a) for a in arrs], sub(phi), sub(ksi), dims, h, 7.5, g=g)
    for a, b in zip(part, whole):
        assert bit_same(a[z_lo - z_base:z_hi - z_base, :H, :W], b[z_lo:z_hi, :H, :W])
    med = oracle.median(arrs[2], dims, 5)
    g5 = oracle.Geom(cdims[1], cdims[0], 2, 4, 9)
    part = oracle.median(np.ascontiguousarray(arrs[2][2:11]), dims, 5, g=g5)
    assert bit_same(part[2:7, :H, :W], med[4:9, :H, :W])


# ---- the oracle's warp and flow statistics against the REFERENCE's own host implementations --------------------------
# (cuda_operation_register_p.cpp:96-139 and cuda_operation_stat_p.cpp:85-104, compiled in place into oracle/_ref; on the
# GPU box the prebuilt library travels with the snapshot, elsewhere these tests skip)

def _need_ref_ops(oracle):
    if not oracle.ref_host_ops():
        pytest.skip("oracle/_ref/libf3d_ref_ops.so not built (no /root/reference here)")


@pytest.mark.parametrize("dims,h", [((37, 21, 9), (1.0, 1.0, 1.0)), ((16, 9, 7), (7.1, 1.6, 1.25)), ((5, 4, 4), (1.3, 0.9, 2.0)),
                                     ((50, 33, 23), (1.05, 1.1, 0.97))])
def test_warp_equals_the_reference_cpu_warp(oracle, dims, h):
    """A.2 pinned: same bits as the reference's plain-C++ warp, including targets outside the volume (~45 % here), exact
    integer landings and NaN flow."""
    _need_ref_ops(oracle)
    W, H, D = dims
    rng = np.random.default_rng(hash(dims) % 2**32)
    f0 = rng.uniform(0, 255, (D, H, W)).astype(np.float32)
    f1 = rng.uniform(0, 255, (D, H, W)).astype(np.float32)
    u, v, w = (rng.uniform(-6, 6, (D, H, W)).astype(np.float32) for _ in range(3))
    u[::3, ::2, ::5] = np.round(u[::3, ::2, ::5])          # exact integer landings (h = 1 case)
    v[1::4, 1::3, 2::7] = np.nan
    w[0, 0, :2] = np.inf
    exp = oracle.ref_warp(f0, f1, u, v, w, h)
    got = oracle.warp(f0, f1, u, v, w, dims, h)
    assert np.array_equal(got.view(np.uint32), exp.view(np.uint32))
    outside = np.count_nonzero(got == f0) / got.size
    assert 0.2 < outside < 0.99      # both branches of the bounds test are exercised


def test_flow_stats_equal_the_reference(oracle):
    _need_ref_ops(oracle)
    rng = np.random.default_rng(4)
    for dims in [(37, 21, 9), (64, 8, 5), (70, 70, 70)]:
        W, H, D = dims
        u, v, w = (rng.uniform(-4, 4, (D, H, W)).astype(np.float32) for _ in range(3))
        mn, mx, avg, _ = oracle.flow_stats(u, v, w, dims)
        assert (mn, mx, avg) == oracle.ref_flow_stats(u, v, w)


def test_residual_stats_match_numpy(oracle):
    """orc_residual_stats (the checker of f3d_residual_stats): double sums and the float maximum of warped - frame_0"""
    rng = np.random.default_rng(8)
    for dims in [(37, 21, 9), (64, 8, 5)]:
        W, H, D = dims
        a, b = (rng.uniform(0, 255, (D, H, W)).astype(np.float32) for _ in range(2))
        ssq, sab, mx = oracle.residual_stats(a, b, dims)
        d = (b - a)
        assert mx == np.abs(d).max()
        assert abs(ssq - (d.astype(np.float64) ** 2).sum()) <= 1e-12 * ssq
        assert abs(sab - np.abs(d).astype(np.float64).sum()) <= 1e-12 * sab


def test_the_reference_kernels_are_built_from_the_reference_tree():
    """oracle/_ref/*.hsaco (tests/ref_kernels.py, tests/test_gpu_reference_kernels.py): code objects for gfx950 holding the
    reference's kernels under the reference's names, made by oracle/Makefile from the .cu files where they lie -- here, where
    /root/reference exists, by running the recipe; on a box without the tree the prebuilt files are looked at as they are."""
    import subprocess
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import ref_kernels
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if os.path.isdir("/root/reference/src/kernels") and os.path.exists("/opt/rocm/bin/hipcc"):
        subprocess.run(["make", "-C", os.path.join(root, "oracle"), "ref"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        assert ref_kernels.available()
    elif not ref_kernels.available():
        pytest.skip("oracle/_ref/*.hsaco not built (no /root/reference here)")
    names = {"solve_3d": [b"compute_phi_ksi_3d", b"solve_3d"], "median_3d": [b"median_3d"], "registration_3d": [b"registration_3d"],
             "resample_3d": [b"resample_x_3d", b"resample_y_3d", b"resample_z_3d"],
             "convolution_3d": [b"convolutionRowsKernel", b"convolutionColumnsKernel", b"convolutionSlicesKernel", b"c_Kernel"],
             "add_3d": [b"add_3d"]}
    for module, symbols in names.items():
        blob = open(os.path.join(ref_kernels.REF_DIR, module + ".hsaco"), "rb").read()
        assert blob[:4] == b"\x7fELF" or blob.startswith(b"__CLANG_OFFLOAD_BUNDLE__"), module   # a code object or a bundle of one
        assert b"gfx950" in blob, module
        for s in symbols + [b"container_size"]:
            assert s in blob, (module, s)
    # nothing of the reference's text is kept in the repository: the recipe names the sources by path
    recipe = open(os.path.join(root, "oracle", "Makefile")).read()
    assert "$(REF)/src/kernels/%.cu" in recipe and "-D__DEVICE_LAUNCH_PARAMETERS_H__" in recipe


def test_the_reference_code_objects_are_what_the_committed_recipe_gives(tmp_path):
    """tests/golden/ref_hsaco_manifest.json pins the MACHINE CODE of the reference kernels the GPU tests load: the sha256 of the
    .text section of the gfx950 code object (the bundle around it embeds build paths; tests/hsaco_text.py).  Where the reference
    tree exists the six files are rebuilt from it into a scratch directory with the committed recipe (oracle/Makefile, HERE
    redirected) and must give the manifest's hashes -- so must the prebuilt files that travel to the GPU box."""
    import subprocess
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import hsaco_text
    import ref_kernels
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    want = {k: v for k, v in hsaco_text.manifest().items() if not k.startswith("_")}
    assert sorted(want) == sorted(m + ".hsaco" for m in ref_kernels.MODULES)
    if os.path.isdir("/root/reference/src/kernels") and os.path.exists("/opt/rocm/bin/hipcc"):
        scratch = str(tmp_path) + "/"
        targets = [scratch + "_ref/" + name for name in want]
        subprocess.run(["make", "-f", os.path.join(root, "oracle", "Makefile"), "HERE=" + scratch] + targets, check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        for name, digest in want.items():
            assert hsaco_text.text_sha256(scratch + "_ref/" + name) == digest, f"{name}: the recipe no longer gives the pinned code"
    elif not ref_kernels.available():
        pytest.skip("no /root/reference here and no prebuilt oracle/_ref")
    if ref_kernels.available():
        for name, digest in want.items():
            assert hsaco_text.text_sha256(os.path.join(ref_kernels.REF_DIR, name)) == digest, f"oracle/_ref/{name} is not the pinned build"
