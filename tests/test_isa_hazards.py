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
