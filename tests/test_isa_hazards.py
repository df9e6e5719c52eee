"""Build-time check of the hand-scheduled solver kernels: no instruction may read the destination register of a hand-issued
load before the wait that covers it, and none of them may spill (tools/isa_hazards.py explains why the compiler cannot know).
Runs on the CPU: hipcc only cross-compiles f3d_solve.hip to assembly."""
import importlib.util
import os
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_tool():
    spec = importlib.util.spec_from_file_location("isa_hazards", os.path.join(ROOT, "tools", "isa_hazards.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="needs hipcc")
def test_no_read_of_a_load_destination_before_its_wait():
    tool = load_tool()
    report, scratch = tool.run()
    names = " ".join(report)
    for kernel in ("k_sweep6", "k_phiksi6", "k_sweep7"):
        assert kernel in names, f"{kernel} not found in the generated assembly"
    for name, bad in report.items():
        assert not bad, f"{name}: {bad[:5]}"
    for name, size in scratch.items():
        assert size == 0, f"{name} uses {size} bytes of scratch per lane"


def test_the_scanner_sees_a_premature_copy():
    """The checker itself: a register copy between a load and its wait is reported, the same copy behind the wait is not."""
    tool = load_tool()
    early = ["global_load_dword v30, v2, s[52:53]", "v_mul_f32_e32 v1, v2, v3", "v_mov_b32_e32 v15, v30", "s_waitcnt vmcnt(0)",
             "v_add_f32_e32 v4, v15, v15", "s_endpgm"]
    late = ["global_load_dword v30, v2, s[52:53]", "v_mul_f32_e32 v1, v2, v3", "s_waitcnt vmcnt(11)", "v_mov_b32_e32 v15, v30",
            "v_add_f32_e32 v4, v15, v15", "s_endpgm"]
    behind_a_branch = ["global_load_dword v30, v2, s[52:53]", "s_waitcnt vmcnt(11)", "s_cbranch_vccnz .LBB0_2", "v_mov_b32_e32 v15, v30",
                       "s_endpgm"]
    assert tool.check_kernel(early)
    assert not tool.check_kernel(late)
    assert tool.check_kernel(behind_a_branch)


# ---- k_pair8: the loader wave's DMA instructions, its counted wait and the barrier that publishes a plane --------------------

# (MODE, TY, FD, YM): two sweeps / sweep + phi/ksi on 4-, 8-, 12-row tiles; on frame derivatives; marching along y (thin volumes)
SHIPPED_PAIR8 = ([(mode, ty, 0, 0) for mode in (0, 1) for ty in (4, 8, 12)] + [(mode, ty, 1, 0) for mode in (0, 1) for ty in (4, 8, 12)] +
                 [(mode, ty, 0, 1) for mode in (0, 1) for ty in (4, 5, 8)])


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="needs hipcc")
def test_pair8_loader_waits_barriers_and_scratch(tmp_path):
    """Every shipped k_pair8 instantiation (two sweeps / sweep + phi/ksi x 4, 8, 12 rows, and the frame-derivative builds): no
    scratch, s_nop in front of every DMA instruction, the counted wait equals the pieces per plane computed HERE from the tile
    shape (and fits vmcnt's six bits), exactly one or two planes are issued between a barrier and that wait, nothing is
    published before it has landed, no LDS read before the first barrier, stores drained before the end.  Then the same check
    must FAIL on a build whose counted wait is off by one -- the round-1 class of bug, caught at build time."""
    tool = load_tool()
    asm = tool.compile_to_asm()
    report, scratch = tool.run_pair8(asm)
    found = {tool.pair8_params(name)[:2] + tool.pair8_params(name)[3:] for name in report if tool.pair8t_params(name) is None}
    assert found == set(SHIPPED_PAIR8), found
    # the tiles without halo rows (k_pair8t: 4 and 5 planes marched along y, two workgroups per CU)
    assert {tool.pair8t_params(name) for name in report if tool.pair8t_params(name)} == {(m, ty) for m in (0, 1) for ty in (4, 5)}
    assert tool.pair8t_per_plane(4) == 12 and tool.pair8t_per_plane(5) == 22
    for name, bad in report.items():
        assert not bad, f"{name}: {bad[:5]}"
    assert len(scratch) >= len(SHIPPED_PAIR8) + 4
    for name, size in scratch.items():
        assert size == 0, f"{name} uses {size} bytes of scratch per lane"
    for mode, ty, fd, ym in SHIPPED_PAIR8:
        assert tool.pair8_per_plane(ty, fd) <= 63
    assert tool.pair8_per_plane(12, 0) == 45 and tool.pair8_per_plane(8, 0) == 34 and tool.pair8_per_plane(4, 0) == 23
    assert tool.pair8_per_plane(12, 1) == 32 and tool.pair8_centre_per_plane(12, 1) == 22 and tool.pair8_centre_per_plane(12, 0) == 0

    # k_tri (three stages per launch, csrc/f3d_solve_tri.h): the same loader idiom, so the same rules -- every instantiation ships
    report3, scratch3 = tool.run_tri(asm)
    assert {tool.tri_params(name) for name in report3} == {(mode, ty) for mode in (0, 1) for ty in (4, 7)}, list(report3)
    for name, bad in report3.items():
        assert not bad, f"{name}: {bad[:5]}"
    assert len(scratch3) == 4 and all(size == 0 for size in scratch3.values()), scratch3
    assert tool.tri_per_plane(4) == 30 and tool.tri_per_plane(7) == 40

    # the mutant: the same sources with the steady-state wait one short
    src_dir = os.path.join(ROOT, "cuda-flow3d_amd", "csrc")
    for f in os.listdir(src_dir):
        shutil.copy(os.path.join(src_dir, f), tmp_path / f)
    for name in ("f3d_solve_pair8.h", "f3d_solve_tri.h"):
        header = tmp_path / name
        text = header.read_text()
        assert text.count('"n"(L::kPerPlane)') == 1
        header.write_text(text.replace('"n"(L::kPerPlane)', '"n"(L::kPerPlane - 1)'))
    import subprocess
    flags = [f if not f.startswith("-I" + src_dir) else "-I" + str(tmp_path) for f in tool.FLAGS]
    out = tmp_path / "mutant.s"
    subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + flags + [str(tmp_path / "f3d_solve.hip"), "-o", str(out)],
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    mutant, _ = tool.run_pair8(str(out))
    assert mutant and all(any(v[0] == "P3" for v in bad) for bad in mutant.values()), "a wrong wait count went unnoticed"
    mutant3, _ = tool.run_tri(str(out))
    assert mutant3 and all(any(v[0] == "P3" for v in bad) for bad in mutant3.values()), "a wrong wait count in k_tri went unnoticed"


def _loader(pieces, wait, nop="s_nop 4", extra=(), barrier_before_wait=False, read_early=False):
    """a toy kernel with the shape of k_pair8: prologue (issue, full wait, barrier), one steady step (barrier, issue, counted wait),
    a compute arm that reads LDS behind the barrier and stores"""
    body = ["s_cmp_eq_u32 s2, 0", "s_cbranch_scc1 .LBB0_9"]
    body += ["ds_read_b32 v9, v8"] if read_early else []
    for _ in range(pieces):
        body += ["s_mov_b32 m0, 0x400", nop, "global_load_lds_dwordx4 v1, s[4:5]"]
    body += ["s_waitcnt vmcnt(0)", "s_barrier", ".LBB0_1:", "s_barrier"]
    for _ in range(pieces):
        body += ["s_mov_b32 m0, 0x800", nop, "global_load_lds_dwordx4 v1, s[4:5]"]
    body += list(extra)
    body += ["s_barrier"] if barrier_before_wait else []
    body += [f"s_waitcnt vmcnt({wait})", "s_add_i32 s3, s3, 1", "s_cmp_lt_i32 s3, s6", "s_cbranch_scc1 .LBB0_1", "s_waitcnt vmcnt(0)", "s_endpgm",
             ".LBB0_9:", "s_barrier", "ds_read_b32 v2, v3", "s_nop 4", "global_store_dword v4, v2, s[8:9]", "s_waitcnt vmcnt(0)", "s_endpgm"]
    return body


def test_the_scanner_sees_a_broken_loader():
    tool = load_tool()
    rules = lambda body, n: sorted({v[0] for v in tool.check_pair8(body, n)})
    assert rules(_loader(5, 5), 5) == []
    assert rules(_loader(5, 4), 5) == ["P3"]                                  # the wait count is not the pieces per plane
    assert rules(_loader(5, 5), 6) == ["P3"]                                  # ... seen from the other side: a piece was dropped
    assert rules(_loader(5, 5, nop="s_nop 1"), 5) == ["P2"]                   # SGPR base right behind a possible VALU write of it
    assert rules(_loader(5, 5, nop="v_mov_b32_e32 v7, v7"), 5) == ["P2"]      # no s_nop between the M0 write and the DMA
    assert rules(_loader(5, 5, extra=["global_load_dword v5, v6, s[4:5]"]), 5) == ["P3"]   # vmcnt would count the extra load
    assert "P4" in rules(_loader(5, 5, barrier_before_wait=True), 5)          # a plane published before its wait
    assert rules(_loader(5, 5, read_early=True), 5) == ["P5"]                 # LDS read before the prologue barrier
    assert rules(_loader(70, 70), 70) == ["P3"]                               # more pieces per plane than vmcnt can count
    body = _loader(5, 5)
    body[-2] = "s_nop 0"                                                      # the stores are not drained before s_endpgm
    assert rules(body, 5) == ["P6"]
