#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native 3-D optical-flow solver.

Metric (BASELINE.json): Mvoxels/s of a FULL coarse-to-fine pyramid solve (default parameters of the reference,
src/main.cpp:77-85) of a 512^3 float32 pair (BASELINE config 4, the config the metric is quoted on).  A "step" is one
ComputeFlow over the whole pyramid (40 levels x (40 x (phi/ksi + 5 sweeps)) + warp, resample, add, median).

`value` is the DEVICE-RESIDENT rate: both frames are already in HBM when the timed region starts and the flow stays there.
The reference's own timer sits before the upload and after the download (optical_flow_e.cpp:169,579); that rate is
reported beside it as `host_inclusive` (page-locked host volumes -> H2D -> solve -> D2H) and is never `value`.

One JSON line on stdout carries the metric plus
  roofline     : the dominant kernel (the fused pair of solver sweeps, 2 x 52 algorithmic B/voxel) timed live with HIP
                 events on the library stream over the timed region, against the 8 TB/s HBM peak; `traffic` is the HBM
                 byte count of one finest-level launch from the committed PMC passes (omitted when they were collected on
                 different kernel sources), `hbm_frac` the same launch priced with those measured bytes;
  fixed_sample : SURVEY.md 8(d)'s like-for-like sample -- phi/ksi + 5 sweeps on the finest level -- on the GPU and on the
                 box's host cores, in Mvoxel-updates/s;
  configs      : BASELINE configs 2 and 3 (the shipped 128^3 pair and the 584x388x5 thin slab, full defaults) with their own
                 roofline fractions;
  cpu_baseline : the same numerics (the oracle, a scalar-per-voxel C port with OpenMP over planes) run on the box's own
                 host cores on BASELINE config 2 (128^3, full default pyramid), timed fully.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--size S] [--no-cpu] [--no-extra]
For N > 1 the SAME volume (512^3 by default, like N = 1: one size per sweep) is z-slab partitioned over one rank per GPU; the line
then carries both exchange orders timed in this one invocation (`exchange_orders`; `value` = the faster), the measured microseconds
per exchange (`exchange_us`), the unsplit single-GPU solve of the same volume on rank 0's device (`single_gpu_same_size`) and the
`speedup` over it, and -- unless --no-config5 / --no-extra -- the same record for BASELINE config 5 (1024^3) under `config5`.  Two ways
in: under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` (RANK / WORLD_SIZE in the environment), or
bare -- `python bench.py --gpus N` starts the N rank processes itself (launch_ranks: the parent never loads the native
library or touches the GPU, forwards rank 0's JSON line, and exits non-zero if any rank fails or hangs).
F3D_SLAB_EXCHANGE=stage|outer in the environment pins one halo-exchange order instead of timing both.
"""
import argparse
import ctypes as C
import hashlib
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SWEEP_BYTES_PER_VOXEL = 52.0    # 10 reads + 3 writes of float32
PHI_KSI_BYTES_PER_VOXEL = 40.0  # 8 reads + 2 writes
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8.0 TB/s spec
# algorithmic bytes of a whole default solve per config
TOTAL_BYTES = {"c2": 0.183e12, "c3": 0.0845e12, 512: 11.58e12, 1024: 92.4e12}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


SOLVER_KERNELS = ("k_pair8", "k_sweep6", "k_sweep7", "k_phiksi6")


def solver_kernel_stamp(library=None):
    """sha256 (16 hex digits) over the MACHINE CODE of the solver kernels in the library being timed: every function symbol of the
    gfx950 code objects inside libf3d_hip.so whose name holds one of SOLVER_KERNELS, name and bytes, sorted by name.  The counter
    record (profiles/*_pmc_traffic.json) is only valid for the kernels it was taken on; rounds 1-3 stamped the SOURCE files, and a
    lab-only edit of a kernel template (a timing probe that the shipped instantiations do not contain) invalidated the record of
    round 3 although no shipped instruction had changed.  What ran is what is hashed now: an edit that leaves the shipped kernels'
    code alone leaves the stamp alone.  Pure Python (clang offload bundle -> ELF64 symbol table), no tool needed on the GPU box.
    tools/pmc_traffic.sh stamps its record with this function; tests/test_abi.py holds the newest record to it."""
    import struct
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    path = library or os.path.join(os.environ.get("F3D_LIBDIR") or os.path.join(ROOT, "cuda-flow3d_amd", "lib"), "libf3d_hip.so")
    with open(path, "rb") as f:
        blob = f.read()
    found = []
    pos = 0
    while True:
        at = blob.find(magic, pos)
        if at < 0:
            break
        pos = at + len(magic)
        (count,) = struct.unpack_from("<Q", blob, at + len(magic))
        if not 0 < count < 16:
            continue
        p = at + len(magic) + 8
        for _ in range(count):
            offset, size, id_len = struct.unpack_from("<QQQ", blob, p)
            p += 24
            triple = blob[p:p + id_len]
            p += id_len
            if not (triple.startswith(b"hip") and triple.endswith(b"gfx950") and size):
                continue
            elf = blob[at + offset:at + offset + size]
            shoff, = struct.unpack_from("<Q", elf, 0x28)
            shentsize, shnum, _ = struct.unpack_from("<HHH", elf, 0x3A)
            sections = [struct.unpack_from("<IIQQQQIIQQ", elf, shoff + i * shentsize) for i in range(shnum)]
            for sec in sections:
                if sec[1] != 2:                                   # SHT_SYMTAB
                    continue
                strings = sections[sec[6]][4]
                for k in range(sec[5] // 24):
                    name, info, _, shndx, value, nbytes = struct.unpack_from("<IBBHQQ", elf, sec[4] + k * 24)
                    if info & 0xf != 2 or not nbytes or shndx >= shnum:   # STT_FUNC with a body
                        continue
                    text = elf[strings + name:elf.index(b"\0", strings + name)].decode()
                    if any(key in text for key in SOLVER_KERNELS):
                        home = sections[shndx]
                        start = home[4] + value - home[3]
                        found.append((text, elf[start:start + nbytes]))
    if not found:
        return None
    h = hashlib.sha256()
    for text, code in sorted(found):
        h.update(text.encode())
        h.update(code)
    return h.hexdigest()[:16]


def measured_traffic(kernel):
    """HBM bytes per finest-level launch of the dominant kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE
    and WRITE_SIZE in separate runs, tools/pmc_traffic.sh; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for
    gfx950).  bench.py cannot collect counters itself; the record carries the stamp of the kernel source it was measured
    on and is dropped when that is not the source of the library being timed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))
    if not files:
        return None
    try:
        doc = json.load(open(files[-1]))
    except (OSError, ValueError):
        return None
    rec = doc.get(kernel)
    if not rec:
        return None
    if doc.get("_solver_kernels_sha16") != solver_kernel_stamp():
        return {"traffic": None, "traffic_scope": f"{os.path.basename(files[-1])} was collected on other solver kernels "
                                                  f"({doc.get('_solver_kernels_sha16')}); not used"}
    return {"traffic": rec["hbm_bytes_per_launch"],
            "traffic_scope": f"one {doc.get('_size', 512)}^3 launch of {kernel}; algorithmic bytes of that launch: "
                             f"{rec['algorithmic_bytes_per_launch']}; source {os.path.basename(files[-1])}",
            "_size": doc.get("_size", 512)}


def golden_pairs():
    """BASELINE configs 2 and 3: the reference's shipped volumes (uint8 fixtures under tests/golden/), as float32"""
    import numpy as np
    g = os.path.join(ROOT, "tests", "golden")
    out = {}
    try:
        a = np.load(os.path.join(g, "inputs_128.npz"))
        out["c2"] = (a["frame_0"].astype(np.float32), a["frame_1"].astype(np.float32))
        r = np.load(os.path.join(g, "inputs_rub.npz"))
        rep = lambda k: np.ascontiguousarray(np.repeat(r[k][None], int(r["depth"]), axis=0).astype(np.float32))
        out["c3"] = (rep("slice_0"), rep("slice_1"))
    except (OSError, KeyError) as e:
        log(f"[bench] golden inputs unavailable: {e}")
    return out


def cpu_baseline(pairs):
    """BASELINE config 2 timed fully on the host: the oracle (CPU port of the same numerics) on the shipped 128^3 pair, full
    default pyramid, every core of the box's share."""
    from oracle import oracle as orc  # cpu_baseline leg only
    if "c2" not in pairs:
        return None
    f0, f1 = pairs["c2"]
    t0 = time.perf_counter()
    orc.compute_flow(f0, f1)
    dt = time.perf_counter() - t0
    return {
        "value": round(f0.size / dt / 1e6, 5), "unit": "Mvoxels/s", "cores": orc.num_threads(), "kind": "port",
        "sample": f"BASELINE config 2 in full: the shipped 128^3 pair, default pyramid (40 levels x 40 x 5), "
                  f"oracle/f3d_oracle.c with OpenMP over planes ({dt:.1f} s)",
    }


def fixed_sample_cpu(S):
    """phi/ksi + 5 sweeps on the finest level of the bench volume, on the host (oracle), in Mvoxel-updates/s
    (one update = one voxel through one of the six kernel passes)."""
    import numpy as np
    pkg = importlib.import_module("cuda-flow3d_amd")
    from oracle import oracle as orc  # cpu_baseline leg only
    f0, f1 = pkg.synth_pair(S, S, S)
    z = lambda: np.zeros((S, S, S), np.float32)
    u, v, w, du, dv, dw = z(), z(), z(), z(), z(), z()
    dims, h = (S, S, S), (1.0, 1.0, 1.0)
    t0 = time.perf_counter()
    phi, ksi = orc.phi_ksi(f0, f1, u, v, w, du, dv, dw, dims, h, 0.001, 0.001)
    for _ in range(5):
        du, dv, dw = orc.solve_sweep(f0, f1, u, v, w, du, dv, dw, phi, ksi, dims, h, 7.5)
    dt = time.perf_counter() - t0
    return {"value": round(6 * S ** 3 / dt / 1e6, 2), "unit": "Mvoxel-updates/s", "cores": orc.num_threads(),
            "seconds": round(dt, 2)}


def fixed_sample_gpu(pkg, S, reps=5):
    """the same six passes through the solve operator (outer 1 x inner 5: du, dv, dw cleared, phi/ksi, five sweeps)"""
    import numpy as np
    f0, f1 = pkg.synth_pair(S, S, S)
    cont = pkg.Containers(S, S, S)
    ptr = {k: cont.new(fill=0) for k in ("f0", "f1", "u", "v", "w", "du", "dv", "dw", "phi", "ksi", "tdu", "tdv", "tdw")}
    cont.upload(ptr["f0"], f0)
    cont.upload(ptr["f1"], f1)
    del f0, f1
    op = pkg.Operation("solve")
    op.initialize(cont)
    kw = dict(dev_frame_0=ptr["f0"], dev_frame_1=ptr["f1"], dev_flow_u=ptr["u"], dev_flow_v=ptr["v"], dev_flow_w=ptr["w"],
              dev_flow_du=ptr["du"], dev_flow_dv=ptr["dv"], dev_flow_dw=ptr["dw"], dev_phi=ptr["phi"], dev_ksi=ptr["ksi"],
              dev_temp_du=ptr["tdu"], dev_temp_dv=ptr["tdv"], dev_temp_dw=ptr["tdw"], outer_iterations_count=1,
              inner_iterations_count=5, equation_alpha=7.5, equation_smoothness=0.001, equation_data=0.001, hx=1.0, hy=1.0,
              hz=1.0, data_size=(S, S, S))
    op.execute(**kw)
    pkg.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        op.execute(**kw)
    pkg.sync()
    dt = (time.perf_counter() - t0) / reps
    op.destroy()
    cont.free()
    return {"value": round(6 * S ** 3 / dt / 1e6, 1), "unit": "Mvoxel-updates/s", "ms": round(dt * 1e3, 3)}


def small_configs(pkg, pairs, reps=5):
    """BASELINE configs 2 and 3 on the GPU, full defaults: ms per solve, Mvoxels/s and the fraction of the HBM roofline
    their algorithmic bytes amount to"""
    names = {"c2": "128^3 shipped pair, full default pyramid (BASELINE config 2)",
             "c3": "584x388x5 thin-slab pair, full default pyramid (BASELINE config 3)"}
    out = []
    for key in ("c2", "c3"):
        if key not in pairs:
            continue
        f0, f1 = pairs[key]
        d, h, w = f0.shape
        flow = pkg.OpticalFlow()
        flow.initialize(w, h, d)
        flow.upload(f0, f1)
        flow.compute_resident(silent=True)
        pkg.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            flow.compute_resident(silent=True)
        pkg.sync()
        dt = (time.perf_counter() - t0) / reps
        flow.destroy()
        gbs = TOTAL_BYTES[key] / dt / 1e9
        out.append({"workload": names[key], "ms_per_step": round(dt * 1e3, 3), "value": round(f0.size / dt / 1e6, 3),
                    "unit": "Mvoxels/s", "roofline_frac": round(gbs / HBM_PEAK_GBS, 4), "achieved_GBs": round(gbs, 1)})
    return out


def host_inclusive(pkg, flow, S, reps=2):
    """the reference's timer placement (optical_flow_e.cpp:169,579): upload of both frames, the solve, download of u, v, w --
    from page-locked host volumes (the reference's ALLOCATE_PINNED_MEMORY switch, data3d.cpp:30,57-61)"""
    import numpy as np
    hip = pkg.hip()
    f0, f1 = pkg.synth_pair(S, S, S)
    out = tuple(np.zeros((S, S, S), np.float32) for _ in range(3))  # the caller's flow volumes, touched
    pinned = []
    for a in (f0, f1) + out:
        if hip.f3d_host_register(C.c_void_p(a.ctypes.data), a.nbytes) == 0:
            pinned.append(a)
    try:
        flow.compute(f0, f1, silent=True, out=out)
        pkg.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            flow.compute(f0, f1, silent=True, out=out)
        pkg.sync()
        dt = (time.perf_counter() - t0) / reps
    finally:
        for a in pinned:
            hip.f3d_host_unregister(C.c_void_p(a.ctypes.data))
    return {"ms_per_step": round(dt * 1e3, 3), "value": round(S ** 3 / dt / 1e6, 4), "unit": "Mvoxels/s", "steps": reps,
            "note": "timer before H2D, after D2H (the reference's placement); the two frames and the three flow volumes are the "
                    "caller's, page-locked (the reference's ALLOCATE_PINNED_MEMORY switch)"}


def profiler_attached():
    """rocprofv3 preloads its tool library: the f3d_prof_* event bracket is switched off under it (the trace gives the kernel
    times, and counter collection serialises every dispatch anyway)"""
    blob = " ".join(os.environ.get(k, "") for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB"))
    return "rocprof" in blob and os.environ.get("F3D_BENCH_FORCE_EVENTS") != "1"


def golden_plane_digest(size):
    """the committed per-plane digest of the default solve of the size^3 synthetic pair (tests/test_gpu_configs.py pins it
    against the resident, unfused, 8-slab and out-of-core drivers), or None"""
    key = {512: "c4_512_default_plane_sha256", 1024: "c5_1024_default_plane_sha256"}.get(size)
    path = os.path.join(ROOT, "tests", "golden", "config_digests.json")
    if key is None or not os.path.exists(path):
        return None
    with open(path) as f:
        return json.load(f).get(key)


def parity_record(size, digest):
    want = golden_plane_digest(size)
    return {"what": "sha256 over the per-plane sha256 of the (u, v, w) this run left on the device(s) against the committed "
                    "single-GPU result of the same pair (tests/golden/config_digests.json; bit-exact or false).  That result is "
                    "also what the reference's own kernels, compiled for gfx950 from the reference tree, give for this pair "
                    "(tests/test_gpu_reference_kernels.py: test_baseline_config_4 / _5_on_the_reference_kernels)",
            "digest": digest, "golden": want, "match": (digest == want) if want else None}


def run_single(args):
    pkg = importlib.import_module("cuda-flow3d_amd")
    S = args.size
    log(f"[bench] generating the {S}^3 synthetic pair on the host ...")
    f0, f1 = pkg.synth_pair(S, S, S)
    flow = pkg.OpticalFlow()
    flow.initialize(S, S, S)
    flow.upload(f0, f1)
    del f0, f1
    hip = pkg.hip()
    events = not profiler_attached()

    for i in range(args.warmup):
        t = flow.compute_resident(silent=True)
        log(f"[bench] warmup {i}: {t:.3f} s")

    # Timed region: HIP events around every launch of the DOMINANT kernel only (the fused pair, kernel id 2; the single
    # sweep when fusion is off).  Events on all 6400 solver launches of a solve cost ~2 % of it; the other two solver kernels
    # are timed in one extra, untimed step afterwards.
    # kernel ids (include/f3d.h): 0 phi/ksi, 1 one sweep, 2 two fused sweeps, 3 sweep + next phi/ksi fused, 4 three fused sweeps,
    # 5 two sweeps + next phi/ksi fused (the three-stage launches of the small and mid-size levels)
    dominant = 2 if os.environ.get("F3D_FUSED_SWEEPS", "1") != "0" else 1
    hip.f3d_prof_reset()
    hip.f3d_prof_select(1 << dominant)
    hip.f3d_prof_enable(1 if events else 0)
    pkg.sync()
    t0 = time.perf_counter()
    dev_s = 0.0
    for i in range(args.steps):
        dev_s += flow.compute_resident(silent=True)
    pkg.sync()
    wall = time.perf_counter() - t0
    hip.f3d_prof_enable(0)
    if events and not args.no_extra:
        hip.f3d_prof_select(0x3f & ~(1 << dominant))
        hip.f3d_prof_enable(1)
        flow.compute_resident(silent=True)      # untimed: phi/ksi and the other sweep kernel
        pkg.sync()
        hip.f3d_prof_enable(0)
    hip.f3d_prof_select(0x3f)
    extra = args.steps                      # their totals cover one step, the dominant kernel's cover `steps`

    def prof(kernel, min_vox):
        ms, n, vox = C.c_double(), C.c_uint64(), C.c_double()
        pkg.check(hip.f3d_prof_read(kernel, min_vox, C.byref(ms), C.byref(n), C.byref(vox)))
        return ms.value, n.value, vox.value

    s1_ms, s1_n, s1_vox = prof(1, 0)          # single sweeps (the odd fifth of every outer iteration)
    s2_ms, s2_n, s2_vox = prof(2, 0)          # fused pairs: two sweeps of algorithmic work per launch
    f2_ms, f2_n, f2_vox = prof(2, S ** 3)
    f1_ms, f1_n, f1_vox = prof(1, S ** 3)
    pk_ms, pk_n, pk_vox = prof(0, 0)
    sp_ms, sp_n, sp_vox = prof(3, 0)          # last sweep of an outer iteration fused with the next phi/ksi
    spf_ms, spf_n, spf_vox = prof(3, S ** 3)  # ... on the finest level alone
    t3_ms, t3_n, t3_vox = prof(4, 0)          # three fused sweeps (small and mid-size levels)
    tp_ms, tp_n, tp_vox = prof(5, 0)          # two sweeps + next phi/ksi (the same levels)
    # the kernels of the extra step ran once, the dominant one `steps` times: put them on the same footing
    pk_ms, pk_n, pk_vox = pk_ms * extra, pk_n * extra, pk_vox * extra
    sp_ms, sp_n, sp_vox = sp_ms * extra, sp_n * extra, sp_vox * extra
    spf_ms, spf_n, spf_vox = spf_ms * extra, spf_n * extra, spf_vox * extra
    t3_ms, t3_n, t3_vox = t3_ms * extra, t3_n * extra, t3_vox * extra
    tp_ms, tp_n, tp_vox = tp_ms * extra, tp_n * extra, tp_vox * extra
    if dominant == 2:
        s1_ms, s1_n, s1_vox = s1_ms * extra, s1_n * extra, s1_vox * extra
        f1_ms, f1_n, f1_vox = f1_ms * extra, f1_n * extra, f1_vox * extra
    hip.f3d_prof_reset()

    # outside the timed region: the bits of the result against the committed single-GPU digest
    parity = parity_record(S, pkg.combine_plane_digests(pkg.flow_plane_digests(flow.download())))
    log(f"[bench] result digest {parity['digest'][:16]}... matches the committed one: {parity['match']}")
    inclusive = None
    if not args.no_extra:
        log("[bench] host-inclusive step (H2D + solve + D2H) ...")
        inclusive = host_inclusive(pkg, flow, S, reps=min(args.steps, 10))   # as many steps as the timed region (at most 10)
    flow.destroy()

    def gbs(bytes_per_voxel, vox, ms):
        return bytes_per_voxel * vox / (ms * 1e-3) / 1e9 if ms else 0.0

    fused = s2_n > 0
    pair_kernel = "k_pair8" if os.environ.get("F3D_PAIR8", "1") != "0" else "k_sweep7"
    if fused:   # dominant kernel: the fused pair
        fd_note = ", frame derivatives read: f3d_solve_sweep2_fd" if os.environ.get("F3D_FRAME_DERIVATIVES", "1") != "0" and pair_kernel == "k_pair8" else ", f3d_solve_sweep2"
        dom_name, dom_b, dom_ms, dom_n, dom_vox = f"{pair_kernel} (two fused solver sweeps{fd_note})", 2 * SWEEP_BYTES_PER_VOXEL, s2_ms, s2_n, s2_vox
        fin_ms, fin_n, fin_vox = f2_ms, f2_n, f2_vox
    else:
        dom_name, dom_b, dom_ms, dom_n, dom_vox = "k_sweep6 (solver sweep, f3d_solve_sweep)", SWEEP_BYTES_PER_VOXEL, s1_ms, s1_n, s1_vox
        fin_ms, fin_n, fin_vox = f1_ms, f1_n, f1_vox
    achieved = gbs(dom_b, dom_vox, dom_ms)
    finest = gbs(dom_b, fin_vox, fin_ms)
    # every solver launch priced at its algorithmic bytes: 52 B per voxel-sweep, 40 B per voxel of phi/ksi
    all_sweeps = gbs(1.0, SWEEP_BYTES_PER_VOXEL * (s1_vox + 2 * s2_vox + sp_vox + 3 * t3_vox + 2 * tp_vox) +
                     PHI_KSI_BYTES_PER_VOXEL * (sp_vox + pk_vox + tp_vox), s1_ms + s2_ms + sp_ms + pk_ms + t3_ms + tp_ms)
    ms_per_step = wall / args.steps * 1e3
    whole = TOTAL_BYTES.get(S)
    out = {
        "metric": "Mvoxels/s full pyramid solve", "value": round(S ** 3 * args.steps / wall / 1e6, 4),
        "unit": "Mvoxels/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{S}^3 synthetic translated-Gaussian float32 pair (BASELINE config "
                               f"{4 if S == 512 else 5 if S == 1024 else 'size ' + str(S)}), full coarse-to-fine pyramid "
                               "(40 levels x 40 outer x 5 inner, alpha 7.5, median 5^3, Gaussian sigma 2), "
                               "frames resident in HBM (value = device-resident rate; host_inclusive beside it)",
                   "parallelism": "1 GPU"},
        "device_ms_per_step": round(dev_s / args.steps * 1e3, 3),
        "host_inclusive": inclusive,
        "parity": parity,
        "whole_run_roofline_frac": round(whole / (wall / args.steps) / 1e9 / HBM_PEAK_GBS, 4) if whole else None,
        "roofline": {
            "bound": "hbm", "kernel": dom_name,
            "algorithmic_bytes_per_voxel_per_launch": dom_b,
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": None, "launches": dom_n, "avg_launch_us": round(dom_ms / dom_n * 1e3, 3) if dom_n else None,
            "avg_voxels_per_launch": round(dom_vox / dom_n, 1) if dom_n else None,
            "finest_level": {"achieved": round(finest, 1), "frac": round(finest / HBM_PEAK_GBS, 4), "launches": fin_n,
                             "avg_launch_us": round(fin_ms / fin_n * 1e3, 3) if fin_n else None,
                             "note": "algorithmic bytes of TWO sweeps per launch (SURVEY.md 8d) over a launch that moves the bytes of one: "
                                     "this fraction is not bounded by 1; hbm_frac below prices the same launch in the bytes the memory "
                                     "system moved"},
            "all_solver_launches": {"achieved": round(all_sweeps, 1), "frac": round(all_sweeps / HBM_PEAK_GBS, 4),
                                    "note": "every solver launch at its algorithmic bytes (52 B per voxel-sweep, 40 B per voxel of "
                                            "phi/ksi), fused or not; the non-dominant kernels are timed in one extra untimed step"},
            "sweep_phi_ksi": {"kernel": "k_pair8 (last sweep + next phi/ksi, f3d_solve_sweep_phi_ksi)",
                              "achieved": round(gbs(SWEEP_BYTES_PER_VOXEL + PHI_KSI_BYTES_PER_VOXEL, sp_vox, sp_ms), 1),
                              "launches": sp_n,
                              "finest_level": {"achieved": round(gbs(SWEEP_BYTES_PER_VOXEL + PHI_KSI_BYTES_PER_VOXEL, spf_vox, spf_ms), 1),
                                               "frac": round(gbs(SWEEP_BYTES_PER_VOXEL + PHI_KSI_BYTES_PER_VOXEL, spf_vox, spf_ms) / HBM_PEAK_GBS, 4),
                                               "launches": spf_n,
                                               "avg_launch_us": round(spf_ms / spf_n * 1e3, 3) if spf_n else None}},
            "three_stage": {"kernel": "k_tri (three sweeps: f3d_solve_sweep3; two sweeps + next phi/ksi: f3d_solve_sweep2_phi_ksi) on the "
                                      "levels up to ~144^3",
                            "three_sweeps": {"achieved": round(gbs(3 * SWEEP_BYTES_PER_VOXEL, t3_vox, t3_ms), 1), "launches": t3_n,
                                             "avg_launch_us": round(t3_ms / t3_n * 1e3, 3) if t3_n else None},
                            "two_sweeps_phi_ksi": {"achieved": round(gbs(2 * SWEEP_BYTES_PER_VOXEL + PHI_KSI_BYTES_PER_VOXEL, tp_vox, tp_ms), 1),
                                                   "launches": tp_n, "avg_launch_us": round(tp_ms / tp_n * 1e3, 3) if tp_n else None}},
            "single_sweep": {"kernel": "k_sweep6", "achieved": round(gbs(SWEEP_BYTES_PER_VOXEL, s1_vox, s1_ms), 1),
                             "launches": s1_n},
            "phi_ksi": {"kernel": "k_phiksi6", "achieved": round(gbs(PHI_KSI_BYTES_PER_VOXEL, pk_vox, pk_ms), 1),
                        "frac": round(gbs(PHI_KSI_BYTES_PER_VOXEL, pk_vox, pk_ms) / HBM_PEAK_GBS, 4), "launches": pk_n},
        },
    }
    if not events:
        out["roofline"]["note"] = "a profiler is attached: no HIP-event bracket in this run, kernel times come from its trace"
    # the record of the kernel that ran: the resident operator reads frame derivatives (F3D_FRAME_DERIVATIVES, on by default) wherever
    # its four extra volumes fit, i.e. the _fd builds of the fused launches
    fd_on = os.environ.get("F3D_FRAME_DERIVATIVES", "1") != "0" and fused and pair_kernel == "k_pair8"
    traffic = measured_traffic("k_pair8_fd" if fd_on else dom_name.split()[0])
    if traffic:
        tsize = traffic.pop("_size", 512)
        out["roofline"].update(traffic)
        # the same finest-level launch priced with the bytes the memory system really moved (fused sweeps move the bytes of
        # one sweep for the algorithmic work of two, so `frac` above may exceed what a streaming kernel could reach)
        if traffic.get("traffic") and fin_n and tsize == S:
            real = traffic["traffic"] / (fin_ms / fin_n * 1e-3) / 1e9
            out["roofline"]["hbm_frac"] = round(real / HBM_PEAK_GBS, 4)
            out["roofline"]["hbm_GBs"] = round(real, 1)
    if not args.no_extra:
        log("[bench] fixed sample on the GPU (phi/ksi + 5 sweeps, finest level) ...")
        out["fixed_sample"] = {"what": f"phi/ksi + 5 sweeps on the {S}^3 level (SURVEY.md 8d): six kernel passes per voxel",
                               "gpu": fixed_sample_gpu(pkg, S)}
        pairs = golden_pairs()
        log("[bench] BASELINE configs 2 and 3 ...")
        out["configs"] = small_configs(pkg, pairs)
        if not args.no_cpu:
            log("[bench] fixed sample and BASELINE config 2 on the host cores (oracle) ...")
            out["fixed_sample"]["cpu"] = fixed_sample_cpu(S)
            cb = cpu_baseline(pairs)
            if cb:
                out["cpu_baseline"] = cb
    print(json.dumps(out), flush=True)


ORDERS = (("per_outer_iteration", False, "once per outer iteration: K + 1 = 6 planes of du, dv, dw, sweeps on widened windows "
                                          "(thin slabs of small levels: 6n planes once per n <= 4 outer iterations)"),
          ("per_stage", True, "after every solver stage: 2 / 1 / 3 planes, stages on the slab itself"))


def measure_multi(pkg, dist, torch, rank, world, S, steps, warmup, args):
    """One size of the multi-GPU sweep: the SAME S^3 volume on `world` z-slabs (strong scaling), BOTH exchange orders timed in this
    one invocation (bit-identical; which wins is a property of the machine's exchange latency), then -- outside every timed
    region -- one solve per order with HIP events around the exchanges (microseconds per exchange) and, on rank 0's device, the
    unsplit single-GPU solve of the same volume, so that the line carries its own speedup."""
    import numpy as np
    hip = pkg.hip()
    halo = 32  # room for four outer iterations per exchange on thin slabs (24 planes) plus the warp reach
    lo, hi = pkg.plan_owned(S, rank, world)
    zlo, zhi = max(0, lo - halo), min(S, hi + halo)
    if rank == 0:
        log(f"[bench] {world} ranks, {S}^3 volume, rank 0 owns planes [{lo},{hi})")
    f0 = np.empty((S, S, S), np.float32)  # only the slab's pages are ever touched
    f1 = np.empty((S, S, S), np.float32)
    local_max = pkg.synth_planes(S, S, S, zlo, zhi, f0, f1)
    t = torch.tensor([local_max], dtype=torch.float32)
    # the global maximum needs every plane once: each rank also scans its OWN planes (halos overlap, max is idempotent)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    scale = np.float32(255.0) / np.float32(t.item())
    f0[zlo:zhi] *= scale
    f1[zlo:zhi] *= scale

    flow = pkg.SlabOpticalFlow(world, [rank], halo_capacity=halo)
    flow.initialize(S, S, S)
    flow.upload(f0, f1)
    del f0, f1
    env_order = os.environ.get("F3D_SLAB_EXCHANGE")          # an explicit choice in the environment pins the order
    orders = [o for o in ORDERS if env_order is None or (o[1] == (env_order == "stage"))]
    if args.one_order:
        orders = orders[:1]
    measured = {}
    for name, per_stage, what in orders:
        flow.set_exchange_per_stage(per_stage)
        for i in range(warmup):
            tsec = flow.compute_resident()
            if rank == 0:
                log(f"[bench] {S}^3 {name} warmup {i}: {tsec:.3f} s")
        before = pkg.comm_info()
        hip.f3d_prof_reset()
        hip.f3d_prof_select(0x6)   # events on the sweep kernels only (ids 1 and 2): they are what the roofline line reports
        hip.f3d_prof_enable(1)
        pkg.sync()
        dist.barrier()
        t0 = time.perf_counter()
        for i in range(steps):
            flow.compute_resident()
        pkg.sync()
        dist.barrier()
        wall = time.perf_counter() - t0
        hip.f3d_prof_enable(0)
        tw = torch.tensor([wall], dtype=torch.float64)
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        wall = tw.item()
        after = pkg.comm_info()
        tot_ms, tot_bytes, launches = 0.0, 0.0, 0
        per_kernel = {}
        for kid, bpv in ((1, SWEEP_BYTES_PER_VOXEL), (2, 2 * SWEEP_BYTES_PER_VOXEL)):
            ms, n, vox = C.c_double(), C.c_uint64(), C.c_double()
            pkg.check(hip.f3d_prof_read(kid, 0, C.byref(ms), C.byref(n), C.byref(vox)))
            tot_ms += ms.value
            tot_bytes += bpv * vox.value
            launches += n.value
            per_kernel[kid] = (ms.value, n.value, vox.value)
        # outside the timed region: every rank hashes the planes it owns, rank 0 compares the whole with the single-GPU digest
        mine = pkg.flow_plane_digests(flow.download(), lo, hi)
        gathered = [None] * world if rank == 0 else None
        dist.gather_object(mine, gathered, dst=0)
        mine_comm = {"rank": after["rank"], "device": after["device"], "sent_bytes": after["sent_bytes"] - before["sent_bytes"],
                     "exchanges": after["exchanges"] - before["exchanges"]}
        comms = [None] * world if rank == 0 else None
        dist.gather_object(mine_comm, comms, dst=0)
        # one more solve, untimed, with events around every exchange: what an exchange costs on THIS machine
        pkg.comm_timing(True)
        flow.compute_resident()
        timing = pkg.comm_timing_read()
        pkg.comm_timing(False)
        timings = [None] * world if rank == 0 else None
        dist.gather_object(timing, timings, dst=0)
        if rank == 0:
            parity = parity_record(S, pkg.combine_plane_digests([[d for part in gathered for d in part[c]] for c in range(3)]))
            log(f"[bench] {S}^3 {name}: {wall / steps * 1e3:.1f} ms per step; digest {parity['digest'][:16]}... matches the committed "
                f"single-GPU one: {parity['match']}")
            achieved = tot_bytes / (tot_ms * 1e-3) / 1e9 if tot_ms else 0.0
            measured[name] = {
                "exchange_order": what, "ms_per_step": round(wall / steps * 1e3, 3), "value": round(S ** 3 * steps / wall / 1e6, 4),
                "parity": parity, "rank0_solver_sweeps_GBs": round(achieved, 1), "rank0_sweep_launches": launches,
                "_per_kernel": per_kernel,
                "comm": {"per_rank": comms, "halo_GB_sent_per_step": round(sum(c["sent_bytes"] for c in comms) / 1e9 / steps, 4),
                         "exchanges_per_step_rank0": comms[0]["exchanges"] // max(1, steps)},
                # microseconds per exchange, measured (HIP events: pack -> grouped send / recv -> unpack on the library stream; the
                # transfer alone on the stream it was posted to): per rank, from one untimed solve
                "exchange_us": {"rank0": timings[0],
                                "worst_rank_mean_us_blocking": max((x["blocking_exchange"]["mean_us"] or 0.0) for x in timings),
                                "worst_rank_mean_us_transfer_alone": max((x["grouped_send_recv_alone"]["mean_us"] or 0.0) for x in timings)},
            }
    comm = pkg.comm_info()
    flow.destroy()
    # the unsplit solve of the same volume on rank 0's device (the others wait): every multi-GPU line carries its own baseline
    single = None
    if rank == 0 and not args.no_single:
        log(f"[bench] single-GPU solve of the same {S}^3 volume on rank 0's device ...")
        g0, g1 = pkg.synth_pair(S, S, S)
        one = pkg.OpticalFlow()
        one.initialize(S, S, S)
        one.upload(g0, g1)
        del g0, g1
        one.compute_resident(silent=True)
        pkg.sync()
        n1 = max(1, min(steps, 3))
        t0 = time.perf_counter()
        for _ in range(n1):
            one.compute_resident(silent=True)
        pkg.sync()
        dt = (time.perf_counter() - t0) / n1
        digest1 = pkg.combine_plane_digests(pkg.flow_plane_digests(one.download()))
        one.destroy()
        single = {"ms_per_step": round(dt * 1e3, 3), "value": round(S ** 3 / dt / 1e6, 4), "unit": "Mvoxels/s", "steps": n1,
                  "what": f"OpticalFlowE (resident, unsplit) on the same {S}^3 pair on rank 0's device, after the timed regions",
                  "digest_equals_the_slab_runs": all(m["parity"]["digest"] == digest1 for m in measured.values())}
    dist.barrier()
    if rank != 0:
        return None
    best = min(measured, key=lambda k: measured[k]["ms_per_step"])
    b = measured[best]
    whole = TOTAL_BYTES.get(S)
    achieved = b["rank0_solver_sweeps_GBs"]
    roof = {"bound": "hbm", "kernel": "k_pair8 + k_sweep6 (all solver sweeps, 52 B per voxel-sweep) on rank 0's slab incl. widened "
                                      "windows", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None, "launches": b["rank0_sweep_launches"]}
    traffic = measured_traffic("k_pair8")
    ms2, n2, vox2 = b["_per_kernel"][2]
    if traffic and traffic.get("traffic") and n2:
        # the fused launches of this run priced with the bytes per voxel the memory system moved in the committed counter
        # passes (one unsplit launch of the finest level): slab windows re-read two more halo planes per chunk, so this is
        # a lower bound of the real rate
        tsize = traffic.pop("_size", 512)
        per_voxel = traffic["traffic"] / float(tsize) ** 3
        real = per_voxel * vox2 / (ms2 * 1e-3) / 1e9
        roof.update({"traffic": traffic["traffic"], "traffic_scope": traffic["traffic_scope"],
                     "hbm_GBs": round(real, 1), "hbm_frac": round(real / HBM_PEAK_GBS, 4),
                     "hbm_frac_scope": f"the {n2} two-sweep launches of rank 0 at the measured {per_voxel:.1f} B per voxel"})
    for m in measured.values():
        m.pop("_per_kernel")
    shm = os.environ.get("F3D_COMM_BACKEND") == "shm"
    out = {
        "metric": "Mvoxels/s full pyramid solve", "value": b["value"],
        "unit": "Mvoxels/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": b["ms_per_step"], "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic", "parity": b["parity"],
        "config": {"workload": f"{S}^3 synthetic translated-Gaussian float32 pair"
                               + (" (BASELINE config 4: the size the metric is quoted on)" if S == 512 else
                                  " (BASELINE config 5, the z-slab scaling workload)" if S == 1024 else "")
                               + ", full coarse-to-fine pyramid (40 levels x 40 outer x 5 inner, alpha 7.5, median 5^3, Gaussian "
                                 "sigma 2), frames resident in HBM; the SAME volume at every N (strong scaling)",
                   "parallelism": f"z-slab decomposition over {world} GPUs, halo exchange on RCCL, {b['exchange_order']}"
                                  + (" -- REHEARSAL: shared-memory transport, ranks share devices" if shm else "")},
        # both bit-identical exchange orders were timed in this invocation: `value` is the faster one
        "exchange_order": best, "exchange_orders": measured,
        # the unsplit solve of the same volume on rank 0's device and what the N GPUs make of it
        "single_gpu_same_size": single,
        "speedup": round(b["value"] / single["value"], 4) if single else None,
        "launched_by": "bench.py itself (one child process per rank)" if os.environ.get("F3D_BENCH_LAUNCHED") == "1"
                       else "an external launcher (RANK / WORLD_SIZE were set)",
        # the communicator as the transport reports it: `rccl_ranks` is ncclCommCount's answer on rank 0, not WORLD_SIZE
        "rccl_ranks": comm["ranks"], "comm_backend": comm["backend"],
        "comm": b["comm"],
        "exchange_us": b["exchange_us"],
        "whole_run_roofline_frac": round(whole / (b["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS / world, 4) if whole else None,
        "roofline": roof,
    }
    return out


def run_multi(args):
    """One rank per GPU (launched by torch.distributed.run, or by this script itself when started bare): the volume is cut into
    z-slabs, halo planes travel over RCCL (ncclSend/ncclRecv inside libf3d_hip.so); torch.distributed (gloo) only carries the
    128-byte RCCL id, the barriers and the max-over-ranks of the timings.  The sweep is quoted on ONE size for every N -- 512^3, the
    size BASELINE.json's metric names -- and, time allowing, BASELINE config 5 (1024^3) rides along in the same line under
    `config5` with its own single-GPU time and speedup."""
    pkg = importlib.import_module("cuda-flow3d_amd")  # load the native library (and /opt/rocm's HIP) before torch
    rank = int(os.environ["RANK"])
    world = int(os.environ["WORLD_SIZE"])
    local_rank = int(os.environ.get("LOCAL_RANK", rank))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    device = local_rank
    if os.environ.get("F3D_COMM_BACKEND") == "shm":
        # rehearsal on a box with fewer GPUs than ranks: the ranks share the devices and the halos travel through
        # shared memory instead of RCCL (which refuses two ranks on one device); never used for a reported number
        count = C.c_int()
        pkg.check(pkg.hip().f3d_device_count(C.byref(count)), "f3d_device_count")
        device = local_rank % max(1, count.value)
    pkg.check(pkg.hip().f3d_init(device), "f3d_init")
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    box = [pkg.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    pkg.comm_init(box[0], rank, world, device=device)

    out = measure_multi(pkg, dist, torch, rank, world, args.size, args.steps, args.warmup, args)
    if args.size == 512 and not args.no_extra and not args.no_config5:
        # BASELINE config 5 in the same invocation, bounded: one warm-up and two steps per exchange order
        c5 = measure_multi(pkg, dist, torch, rank, world, 1024, min(args.steps, 2), 1, args)
        if rank == 0:
            out["config5"] = c5
    pkg.comm_destroy()
    if rank == 0:
        if not args.no_extra:
            # the legs of the one-GPU line that do not depend on N, so that every line of a scaling run is self-contained: the
            # like-for-like sample on rank 0's device and the host-CPU baseline (the other ranks wait at the barrier below)
            log("[bench] fixed sample on rank 0's GPU (phi/ksi + 5 sweeps on a 512^3 level) ...")
            out["fixed_sample"] = {"what": "phi/ksi + 5 sweeps on a 512^3 level (SURVEY.md 8d): six kernel passes per voxel, "
                                           "on rank 0's device alone", "gpu": fixed_sample_gpu(pkg, 512)}
            if not args.no_cpu:
                log("[bench] fixed sample and BASELINE config 2 on the host cores (oracle) ...")
                out["fixed_sample"]["cpu"] = fixed_sample_cpu(512)
                cb = cpu_baseline(golden_pairs())
                if cb:
                    out["cpu_baseline"] = cb
        print(json.dumps(out), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def launch_ranks(args, argv):
    """`python bench.py --gpus N` started bare: the parent becomes a launcher.  It imports neither the native package nor
    torch and never touches HIP (a process that has initialised the GPU must not be re-executed, and a parent holding a HIP
    context would sit on device 0); it starts N children of this same script with RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT set (child r uses device r), passes rank 0's stdout -- the ONE JSON line -- through, lets the
    children's stderr through as it comes, and waits.  Any child that fails, or the whole job exceeding --launch-timeout,
    ends the others (exact PIDs, SIGTERM then SIGKILL) and makes the exit code non-zero."""
    import signal
    import socket
    import subprocess
    n = args.gpus
    with socket.socket() as sock:       # a free rendezvous port for gloo
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "F3D_BENCH_LAUNCHED": "1"})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs between processes on this driver
        out = subprocess.PIPE if r == 0 else subprocess.DEVNULL   # only rank 0 prints the line
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdout=out))
    log(f"[bench] launcher: started {n} ranks (pids {[p.pid for p in procs]}), rendezvous 127.0.0.1:{port}")

    def stop_all():
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t_end = time.time() + 10
        for p in procs:
            try:
                p.wait(max(0.1, t_end - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()

    deadline = time.time() + args.launch_timeout
    line, failed = None, None
    import threading
    box = {}

    def read_rank0():
        box["out"] = procs[0].stdout.read()
    reader = threading.Thread(target=read_rank0, daemon=True)
    reader.start()
    try:
        while True:
            codes = [p.poll() for p in procs]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                failed = f"rank {bad[0][0]} exited with code {bad[0][1]}"
                break
            if all(c == 0 for c in codes):
                break
            if time.time() > deadline:
                failed = f"no result after {args.launch_timeout} s"
                break
            time.sleep(0.2)
    except KeyboardInterrupt:
        failed = "interrupted"
    if failed:
        stop_all()
    reader.join(5)
    text = (box.get("out") or b"").decode(errors="replace")
    for candidate in text.splitlines():
        if candidate.startswith("{"):
            line = candidate
    if failed or line is None:
        log(f"[bench] launcher: {failed or 'rank 0 printed no result line'}")
        return 1
    print(line, flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=0, help="volume edge; default 512 for every N (BASELINE config 4: the size the metric "
                                                      "is quoted on); with several GPUs the 1024^3 leg (config 5) is appended")
    ap.add_argument("--no-cpu", action="store_true", help="skip the host-CPU legs (cpu_baseline, fixed_sample.cpu)")
    ap.add_argument("--no-extra", action="store_true", help="only the timed steps: no host-inclusive step, fixed sample, "
                                                           "configs 2/3 or CPU legs (profiling runs)")
    ap.add_argument("--no-config5", action="store_true", help="--gpus N > 1: do not append the 1024^3 leg (BASELINE config 5) to the 512^3 line")
    ap.add_argument("--no-single", action="store_true", help="--gpus N > 1: skip the single-GPU solve of the same volume on rank 0 "
                                                             "(no speedup in the line)")
    ap.add_argument("--one-order", action="store_true", help="--gpus N > 1: time only the default exchange order")
    ap.add_argument("--launch-timeout", type=int, default=3300, help="--gpus N started bare: seconds the launcher waits for the "
                                                                     "ranks before it ends them")
    args = ap.parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        # started bare (the way `--gpus 1` is): become the launcher -- before anything loads the native library
        sys.exit(launch_ranks(args, sys.argv[1:]))
    # the native side prints its init/progress lines with printf: keep stdout for the ONE JSON line
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    sys.stdout = os.fdopen(real_stdout, "w")
    multi = args.gpus > 1 or int(os.environ.get("WORLD_SIZE", "1")) > 1
    if args.size <= 0:
        args.size = 512   # one size per sweep: the N = 1, 2, 4, 8 lines of a scaling run are lines of the same volume
    force_multi = os.environ.get("F3D_BENCH_FORCE_SLAB") == "1"  # rehearse the multi-GPU code path with one rank
    if args.gpus == 1 and int(os.environ.get("WORLD_SIZE", "1")) == 1 and not force_multi:
        run_single(args)
    else:
        run_multi(args)


if __name__ == "__main__":
    main()
