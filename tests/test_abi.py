"""The C-ABI boundary without a GPU: both shared libraries load, export every symbol the headers declare, fail loudly
when no device is present, and the product never reaches into the oracle."""
import ctypes as C
import glob
import importlib
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(f3d_[a-z0-9_]+)\s*\(", text)))


@pytest.mark.parametrize("header,lib", [("f3d.h", "hip"), ("f3d_host.h", "host")])
def test_every_declared_symbol_is_exported(f3d, header, lib):
    handle = getattr(f3d, lib)()
    names = declared(header)
    assert len(names) > 20
    missing = [n for n in names if not hasattr(handle, n)]
    assert not missing, f"{header} declares symbols the library does not export: {missing}"


def test_no_device_is_a_loud_error_not_a_fallback(f3d):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    hip = f3d.hip()
    assert hip.f3d_init(-1) != 0
    assert hip.f3d_last_error()
    assert hip.f3d_stream_sync() != 0  # every entry point refuses to run before f3d_init
    assert b"f3d_init" in hip.f3d_last_error()
    flow = f3d.OpticalFlow()
    with pytest.raises(f3d.F3dError):
        flow.initialize(16, 16, 16)


def test_missing_library_raises(f3d, monkeypatch, tmp_path):
    monkeypatch.setattr(f3d, "_LIBDIR", str(tmp_path))
    monkeypatch.setattr(f3d, "_hip", None)
    with pytest.raises(f3d.F3dError, match="no fallback"):
        f3d.hip()


def test_product_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "cuda-flow3d_amd")
    sources = [p for ext in ("py", "cpp", "h", "hip") for p in glob.glob(os.path.join(pkg, "**", "*." + ext), recursive=True)]
    sources += [os.path.join(ROOT, "include", h) for h in ("f3d.h", "f3d_host.h")]
    assert len(sources) > 15
    for path in sources:
        text = open(path).read().lower()
        assert "oracle" not in text, f"{path} mentions the oracle"
    # and the built libraries do not link it
    for so in glob.glob(os.path.join(pkg, "lib", "*.so")):
        assert b"liboracle" not in open(so, "rb").read()


def test_struct_layouts_match_the_header(f3d):
    assert C.sizeof(f3d.Size4) == 4 * C.sizeof(C.c_size_t)       # DataSize4: 4 x size_t, 32 bytes
    assert C.sizeof(f3d.Slab) == 12
    p = f3d.FlowParams()
    f3d.host().f3d_flow_default_params(C.byref(p))
    got = {k: getattr(p, k) for k, _ in f3d.FlowParams._fields_}
    for k, v in f3d.DEFAULT_PARAMS.items():
        assert got[k] == pytest.approx(v), k


def test_a_fatal_signal_leaves_the_load_map_and_still_reaches_the_previous_handler(tmp_path):
    """F3D_CRASH_MAPS: /proc/self/maps is written before Python's faulthandler (installed earlier) reports the fault"""
    import subprocess
    import sys
    maps = tmp_path / "maps.txt"
    code = ("import ctypes, faulthandler, importlib, sys\n"
            "faulthandler.enable()\n"
            f"sys.path.insert(0, {ROOT!r})\n"
            "pkg = importlib.import_module('cuda-flow3d_amd')\n"
            "pkg.hip()\n"
            "ctypes.string_at(16)\n")
    p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, F3D_CRASH_MAPS=str(maps)), capture_output=True, timeout=120)
    assert p.returncode != 0
    assert b"Fatal Python error: Segmentation fault" in p.stderr         # the handler that was there before still ran
    text = maps.read_text()
    assert text.startswith("signal 0x000000000000000b\nfault address 0x0000000000000010\n")
    assert "libf3d_hip.so" in text and "[stack]" in text


def test_the_shipped_library_has_no_switch_that_changes_a_result():
    """The timing builds of the solver kernels that skip parts of the work (F3D_ABLATE*: wrong results) are compiled only under
    -DF3D_LAB into lib/lab/ (make lab, tools/kbench.py --ablate); the product library neither reads those variables nor contains
    an ablated instantiation (k_pair8<MODE, TY, ABL != 0, ...>, k_sweep7<TY, ABL != 0>, k_sweep6<ABLATE != 0, ...>)."""
    import subprocess
    so = os.path.join(ROOT, "cuda-flow3d_amd", "lib", "libf3d_hip.so")
    blob = open(so, "rb").read()
    assert b"F3D_ABLATE" not in blob
    names = subprocess.run(["nm", "-C", so], capture_output=True, text=True).stdout
    stubs = set(re.findall(r"__device_stub__(k_(?:pair8|sweep7|sweep6)<[^>]*>)", names))
    assert len(stubs) >= 20, stubs
    for k in stubs:
        args = [a.strip() for a in k[k.index("<") + 1:-1].split(",")]
        abl = {"k_pair8": args[2] if len(args) > 2 else "0", "k_sweep7": args[1] if len(args) > 1 else "0", "k_sweep6": args[0]}[k[:k.index("<")]]
        assert abl == "0", f"{k} is a timing build and must not ship"
    # the sources keep every such switch behind the macro
    for name in ("f3d_solve.hip", "f3d_solve_pair8.h"):
        text = open(os.path.join(ROOT, "cuda-flow3d_amd", "csrc", name)).read()
        outside = re.sub(r"#ifdef F3D_LAB\b.*?#endif", "", text, flags=re.S)
        assert 'getenv("F3D_ABLATE' not in outside, name


def test_the_newest_counter_record_is_of_the_shipped_kernels():
    """bench.py's roofline.traffic comes from the newest profiles/*_pmc_traffic.json and is dropped when that record was collected on
    other kernels (round 3's driver line lost it that way): the newest record must carry the stamp of the solver kernels' machine code
    in the built library (bench.solver_kernel_stamp: a source edit that leaves the shipped instructions alone leaves it alone).
    After a change that moves it: tools/pmc_traffic.sh on the GPU, copy the record."""
    import json
    import sys
    sys.path.insert(0, ROOT)
    import bench
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))
    assert files
    doc = json.load(open(files[-1]))
    assert doc.get("_solver_kernels_sha16") == bench.solver_kernel_stamp(), (
        f"{os.path.basename(files[-1])} was collected on other solver kernels: run tools/pmc_traffic.sh and commit the new record")
    assert bench.measured_traffic("k_pair8")["traffic"]
