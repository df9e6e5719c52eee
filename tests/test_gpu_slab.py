"""z-slab decomposition on ONE GPU: the multi-GPU driver run with every rank in this process (halo planes copied
between the ranks' containers instead of sent over RCCL) must reproduce the single-GPU driver bit for bit -- the sweep
is a Jacobi update, so widened windows and exchanged halos carry exactly the neighbour's values."""
import numpy as np
import pytest

from conftest import same

pytestmark = pytest.mark.gpu


def single(f3d, f0, f1, **kw):
    d, h, w = f0.shape
    flow = f3d.OpticalFlow()
    flow.initialize(w, h, d)
    out = flow.compute(f0, f1, silent=True, **kw)
    flow.destroy()
    return out


def slabbed(f3d, f0, f1, n_ranks, halo_capacity=16, counters=None, **kw):
    d, h, w = f0.shape
    flow = f3d.SlabOpticalFlow(n_ranks, list(range(n_ranks)), halo_capacity=halo_capacity)
    flow.initialize(w, h, d)
    out = flow.compute(f0, f1, **kw)
    if counters is not None:
        counters["batched"] = flow.batched_exchanges()
    flow.destroy()
    return out


@pytest.mark.parametrize("forced,halo,n_ranks", [("1", 16, 4), ("", 16, 4), ("", 32, 8), ("4", 32, 3), ("3", 40, 8)])
def test_outer_iterations_per_exchange(f3d, monkeypatch, forced, halo, n_ranks):
    """Thin slabs of small levels take n (K + 1) halo planes at once and run n outer iterations between exchanges, on nested
    windows: forced to 1 (the plain order), chosen by the rule, forced to 3 and 4 -- always the single-GPU bits."""
    if forced:
        monkeypatch.setenv("F3D_SLAB_OUTER_PER_EXCHANGE", forced)
    else:
        monkeypatch.delenv("F3D_SLAB_OUTER_PER_EXCHANGE", raising=False)
    f0, f1 = f3d.synth_pair(44, 36, 50)
    kw = dict(warp_levels_count=12, outer_iterations_count=7)
    exp = single(f3d, f0, f1, **kw)
    c = {}
    got = slabbed(f3d, f0, f1, n_ranks, halo_capacity=halo, counters=c, **kw)
    for g, e, n in zip(got, exp, "uvw"):
        assert same(g, e), f"{n_ranks} slabs, forced={forced!r}: component {n} differs, max {np.abs(g - e).max():.3e}"
    if forced == "1":
        assert c["batched"] == 0
    elif forced == "4":
        assert c["batched"] == 12 * 2          # 7 outer iterations = 4 + 3 on every level
    elif forced == "3":
        assert c["batched"] == 12 * 2          # 3 + 3 + 1: two groups of more than one
    else:
        assert c["batched"] > 0


@pytest.mark.parametrize("n_ranks", [1, 2, 3, 8])
def test_slabs_equal_single_gpu(f3d, n_ranks):
    f0, f1 = f3d.synth_pair(48, 40, 44)
    kw = dict(warp_levels_count=14, outer_iterations_count=4)
    exp = single(f3d, f0, f1, **kw)
    got = slabbed(f3d, f0, f1, n_ranks, **kw)
    for g, e, n in zip(got, exp, "uvw"):
        assert same(g, e), f"{n_ranks} slabs: component {n} differs, max {np.abs(g - e).max():.3e}"


def test_thin_slabs_full_defaults(f3d):
    """8 ranks on 40 planes: 5 planes per rank at the finest level, fewer than the 6-plane solver halo, and coarse
    levels where some ranks own nothing -- the exchange plan reaches across several ranks."""
    f0, f1 = f3d.synth_pair(40, 36, 40)
    exp = single(f3d, f0, f1)
    got = slabbed(f3d, f0, f1, 8)
    for g, e, n in zip(got, exp, "uvw"):
        assert same(g, e), f"component {n} differs, max {np.abs(g - e).max():.3e}"


def test_anisotropic_and_no_blur(f3d):
    f0, f1 = f3d.synth_pair(70, 33, 26)
    kw = dict(warp_levels_count=9, outer_iterations_count=3, inner_iterations_count=3, gaussian_sigma=0.0, median_radius=3)
    exp = single(f3d, f0, f1, **kw)
    got = slabbed(f3d, f0, f1, 4, **kw)
    for g, e, n in zip(got, exp, "uvw"):
        assert same(g, e), f"component {n} differs"


def test_single_rank_rccl_init_and_allreduce(f3d):
    """librccl loads, a one-rank communicator comes up and the scalar all-reduce round-trips (the N > 1 exchange itself
    needs N GPUs and is exercised by bench.py --gpus N on the multi-GPU node)."""
    import ctypes as C
    f3d.comm_init(f3d.comm_unique_id(), 0, 1)
    v = C.c_float(3.5)
    f3d.check(f3d.hip().f3d_comm_allreduce_max_f32(C.byref(v)))
    assert v.value == 3.5
    f3d.comm_destroy()


@pytest.mark.parametrize("split", [False, True])
def test_exchange_legs_pack_rccl_self_unpack(f3d, split):
    """The three legs of one halo exchange as the slab driver issues them (optical_flow_slab.cpp, Exchange): ONE pack
    launch over several (field, plane range) segments, a grouped ncclSend/ncclRecv -- here to the rank itself, which is
    all one GPU offers -- and ONE unpack launch; the planes must arrive bit for bit where the segment table says."""
    import ctypes as C
    W, H, D = 70, 13, 9
    rng = np.random.default_rng(7)
    vols = [rng.standard_normal((D, H, W)).astype(np.float32) for _ in range(3)]
    sent = [v.copy() for v in vols]
    box = f3d.Containers(W, H, D)
    src = [box.alloc() for _ in range(3)]
    dst = [box.alloc(fill=0xFF) for _ in range(3)]
    box.set_current()
    for p, v in zip(src, vols):
        box.upload(p, v)
    segs = [(0, 1, 3), (1, 0, 2), (2, 6, 3), (0, 7, 1)]           # (field, first plane, planes)
    shift = [4, 5, 0, 2]                                         # destination first plane per segment
    plane = W * H
    total = sum(s[2] for s in segs) * plane
    stage_s, stage_r, pitch = C.c_uint64(), C.c_uint64(), C.c_size_t()
    hip = f3d.hip()
    f3d.check(hip.f3d_alloc_pitched(C.byref(stage_s), C.byref(pitch), total * 4, 1))
    f3d.check(hip.f3d_alloc_pitched(C.byref(stage_r), C.byref(pitch), total * 4, 1))
    n = len(segs)
    dp, ci, sz = C.c_uint64 * n, C.c_int * n, C.c_size_t * n
    offs, pos = [], 0
    for s in segs:
        offs.append(pos)
        pos += s[2] * plane
    f3d.comm_init(f3d.comm_unique_id(), 0, 1)
    try:
        f3d.check(hip.f3d_pack_segments(dp(*[src[s[0]] for s in segs]), ci(*[s[1] for s in segs]), ci(*[s[2] for s in segs]),
                                        sz(*offs), n, W, H, stage_s.value))
        one = C.c_size_t * 1
        if split:   # the transfer on RCCL's side stream, a kernel of the library stream beside it, then the join
            f3d.check(hip.f3d_comm_sendrecv_begin(stage_s.value, one(0), one(total), stage_r.value, one(0), one(total),
                                                  (C.c_int * 1)(0), 1))
            f3d.check(hip.f3d_add(src[0], src[1], W, H, D, None))
            f3d.check(hip.f3d_comm_sendrecv_end())
            vols[0] = vols[0] + vols[1]            # the add ran after the pack: the packed planes are the old ones
        else:
            f3d.check(hip.f3d_comm_sendrecv(stage_s.value, one(0), one(total), stage_r.value, one(0), one(total),
                                            (C.c_int * 1)(0), 1))
        f3d.check(hip.f3d_unpack_segments(dp(*[dst[s[0]] for s in segs]), ci(*shift), ci(*[s[2] for s in segs]),
                                          sz(*offs), n, W, H, stage_r.value))
        f3d.sync()
    finally:
        f3d.comm_destroy()
    got = [box.download(p, (W, H, D)) for p in dst]
    exp = [np.full((D, H, W), np.nan, np.float32) for _ in range(3)]
    for (f, p0, cnt), d0 in zip(segs, shift):
        exp[f][d0:d0 + cnt] = sent[f][p0:p0 + cnt]
    for g, e in zip(got, exp):
        filled = ~np.isnan(e)
        assert np.array_equal(g[filled], e[filled])
        assert np.isnan(g[~filled]).all(), "unpack wrote outside its segments"
    for p in (stage_s.value, stage_r.value):
        f3d.check(hip.f3d_free(p))
    box.free()


@pytest.mark.parametrize("n_ranks,halo", [(4, 16), (8, 16), (8, 8)])
def test_warp_reach_beyond_the_halo_room_gathers_frame_1(f3d, n_ranks, halo):
    """A pair that moves ~10 planes along z on slabs of 16 or 8 planes: the warp of the fine levels reads frame 1 further away than
    the local containers have halo room (and, with 8 ranks, beyond the neighbouring slab).  The driver gathers frame 1 into a
    container of its own from every rank the reach spans and warps from there (SURVEY 8e fallback; it used to stop with "raise
    halo_capacity"): the single-GPU bits."""
    f0, _ = f3d.synth_pair(48, 40, 64)
    f1 = np.ascontiguousarray(np.roll(f0, 10, axis=0))
    kw = dict(outer_iterations_count=6)
    exp = single(f3d, f0, f1, **kw)
    assert np.abs(exp[2]).max() > 8.0, f"the solver recovered only |w| <= {np.abs(exp[2]).max():.2f}: the case does not test the reach"
    d, h, w = f0.shape
    flow = f3d.SlabOpticalFlow(n_ranks, list(range(n_ranks)), halo_capacity=halo)
    flow.initialize(w, h, d)
    got = flow.compute(f0, f1, **kw)
    gathered = flow.gathered_warps()
    flow.destroy()
    assert gathered >= 1
    for g, e, n in zip(got, exp, "uvw"):
        assert same(g, e), f"{n_ranks} slabs, halo {halo}: component {n} differs, max {np.abs(g - e).max():.3e}"


@pytest.mark.parametrize("n_ranks,dims", [(2, (48, 40, 44)), (4, (44, 36, 50)), (8, (40, 36, 64))])
def test_exchange_after_every_solver_stage(f3d, monkeypatch, n_ranks, dims):
    """F3D_SLAB_EXCHANGE=stage: the increments travel after every fused pair / last sweep, as deep as the next stage reads (2, 1, 3
    planes), and every launch runs on the slab itself instead of on windows widened by up to four planes.  The other order of the
    same arithmetic: single-GPU bits, on thick and on thin slabs (8 ranks: slabs thinner than the halo on the coarse levels)."""
    monkeypatch.setenv("F3D_SLAB_EXCHANGE", "stage")
    f0, f1 = f3d.synth_pair(*dims)
    kw = dict(warp_levels_count=12, outer_iterations_count=5)
    exp = single(f3d, f0, f1, **kw)
    d, h, w = f0.shape
    flow = f3d.SlabOpticalFlow(n_ranks, list(range(n_ranks)), halo_capacity=16)
    flow.initialize(w, h, d)
    got = flow.compute(f0, f1, **kw)
    n_stage = flow.stage_exchanges()
    flow.destroy()
    assert n_stage == 12 * (5 * 3 - 1), n_stage
    for g, e, n in zip(got, exp, "uvw"):
        assert same(g, e), f"{n_ranks} slabs exchanging per stage: component {n} differs, max {np.abs(g - e).max():.3e}"
