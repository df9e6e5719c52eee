"""`python bench.py --gpus N` started bare (the way the one-GPU bench is started) launches its own ranks.

CPU part: the launching parent loads neither the native package nor torch nor any HIP library (a process holding a HIP context
must not be the one that spawns the ranks), hands RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* to its children and turns a failing
rank into a non-zero exit code.  GPU part: the real thing with two rank processes on the box's one GPU (shared-memory transport,
because RCCL refuses two ranks on one device): one JSON line, n_gpus == 2, the result bit-identical to a single-GPU solve."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")

# runs bench.py's main() in a process whose imports and dlopens are recorded by an audit hook; the children are fresh
# interpreters (sys.executable bench.py ...) and are not recorded
AUDITED = textwrap.dedent("""
    import json, runpy, sys
    seen = {"import": [], "dlopen": []}
    def hook(event, args):
        if event == "import":
            seen["import"].append(str(args[0]))
        elif event == "ctypes.dlopen":
            seen["dlopen"].append(str(args[0]))
    sys.addaudithook(hook)
    report, bench = sys.argv[1], sys.argv[2]
    sys.argv = [bench] + sys.argv[3:]
    code = 0
    try:
        runpy.run_path(bench, run_name="__main__")
    except SystemExit as e:
        code = e.code if isinstance(e.code, int) else 1
    json.dump({"exit": code, **seen}, open(report, "w"))
    sys.exit(code)
""")


def test_the_launching_parent_stays_clear_of_the_native_library_and_reports_failing_ranks(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the ranks would run the whole benchmark")
    report = tmp_path / "audit.json"
    env = dict(os.environ)
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)
    env["F3D_COMM_BACKEND"] = "shm"
    p = subprocess.run([sys.executable, "-c", AUDITED, str(report), BENCH, "--gpus", "2", "--size", "64", "--steps", "1",
                        "--warmup", "0", "--no-extra", "--launch-timeout", "300"], env=env, capture_output=True, timeout=600)
    err = p.stderr.decode(errors="replace")
    # no GPU here: every rank fails in f3d_init, loudly, and the launcher turns that into its own failure
    assert p.returncode != 0
    assert "launcher: started 2 ranks" in err
    assert "exited with code" in err
    assert p.stdout.strip() == b"", "no JSON line may appear when a rank failed"
    assert "no ROCm-capable device" in err or "no HIP device" in err or "f3d_init" in err, err[-1500:]   # the children got as far as the native library
    seen = json.load(open(report))
    assert seen["exit"] != 0
    loaded = " ".join(seen["import"])
    assert "cuda-flow3d_amd" not in loaded and "torch" not in seen["import"] and "numpy" not in seen["import"], loaded
    assert not [d for d in seen["dlopen"] if "f3d" in d or "hip" in d or "rccl" in d], seen["dlopen"]


def test_ranks_get_the_torch_distributed_environment(tmp_path, monkeypatch):
    """launch_ranks with the process creation replaced: what each child would have been given"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", BENCH)
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    started = []

    class FakeProc:
        def __init__(self, cmd, env=None, stdout=None):
            self.pid = 1000 + len(started)
            self.rank = int(env["RANK"])
            started.append((cmd, env, stdout))
            self.stdout = self
        def read(self):
            return b'native noise\n{"metric": "x", "n_gpus": 3}\n' if self.rank == 0 else b""
        def poll(self):
            return 0
        def wait(self, timeout=None):
            return 0
        def terminate(self):
            pass
        kill = terminate

    import subprocess as sp
    monkeypatch.setattr(sp, "Popen", FakeProc)
    printed = []
    monkeypatch.setattr("builtins.print", lambda *a, **k: printed.append((a, k)))
    args = type("A", (), {"gpus": 3, "launch_timeout": 30})()
    rc = bench.launch_ranks(args, ["--gpus", "3", "--size", "128"])
    assert rc == 0
    assert len(started) == 3
    ports = set()
    for r, (cmd, env, out) in enumerate(started):
        assert cmd[0] == sys.executable and cmd[1] == BENCH and cmd[2:] == ["--gpus", "3", "--size", "128"]
        assert env["RANK"] == str(r) and env["LOCAL_RANK"] == str(r) and env["WORLD_SIZE"] == "3"
        assert env["MASTER_ADDR"] == "127.0.0.1"
        ports.add(env["MASTER_PORT"])
        assert (out == sp.PIPE) == (r == 0)
    assert len(ports) == 1
    lines = [a[0] for a, k in printed if k.get("file") is None]
    assert lines == ['{"metric": "x", "n_gpus": 3}']       # rank 0's JSON line and nothing else on stdout


@pytest.mark.gpu
def test_bench_gpus_2_started_bare_equals_one_gpu(f3d):
    S = 128
    f0, f1 = f3d.synth_pair(S, S, S)
    flow = f3d.OpticalFlow()
    flow.initialize(S, S, S)
    flow.upload(f0, f1)
    flow.compute_resident(silent=True)
    want = f3d.combine_plane_digests(f3d.flow_plane_digests(flow.download()))
    flow.destroy()
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    env["F3D_COMM_BACKEND"] = "shm"       # two ranks on the one GPU of the box; never a reported number
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--size", str(S), "--steps", "1", "--warmup", "0", "--no-extra"],
                       env=env, capture_output=True, timeout=900)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 1 and line["unit"] == "Mvoxels/s"
    assert line["launched_by"].startswith("bench.py itself")
    assert line["comm_backend"] == "shm" and line["rccl_ranks"] == 2
    assert [c["rank"] for c in line["comm"]["per_rank"]] == [0, 1]
    assert all(c["sent_bytes"] > 0 and c["exchanges"] > 0 for c in line["comm"]["per_rank"])
    assert line["parity"]["digest"] == want, "two rank processes did not reproduce the single-GPU bits"
    assert line["value"] == pytest.approx(S ** 3 / (line["ms_per_step"] * 1e-3) / 1e6, rel=1e-3)
    # one size per sweep, and everything a scaling run needs in the one line (round-3 verdict, item 1):
    assert f"{S}^3" in line["config"]["workload"] and line["scaling"] == "strong"
    # ... both exchange orders timed in this invocation, each with the single-GPU bits, `value` the faster of the two
    orders = line["exchange_orders"]
    assert set(orders) == {"per_outer_iteration", "per_stage"}
    for name, rec in orders.items():
        assert rec["parity"]["digest"] == want, name
        assert all(c["sent_bytes"] > 0 and c["exchanges"] > 0 for c in rec["comm"]["per_rank"]), name
        # ... microseconds per exchange, measured with events around pack -> transfer -> unpack on every rank
        blocking = rec["exchange_us"]["rank0"]["blocking_exchange"]
        assert blocking["count"] > 0 and blocking["mean_us"] > 0 and blocking["mean_bytes_sent"] > 0, (name, blocking)
        assert rec["exchange_us"]["worst_rank_mean_us_blocking"] >= blocking["mean_us"] * 0.999
    assert orders["per_stage"]["comm"]["exchanges_per_step_rank0"] > orders["per_outer_iteration"]["comm"]["exchanges_per_step_rank0"]
    assert line["exchange_order"] in orders and line["ms_per_step"] == min(r["ms_per_step"] for r in orders.values())
    # ... the unsplit solve of the same volume on rank 0's device, its bits, and the speedup computed from it
    single = line["single_gpu_same_size"]
    assert single["digest_equals_the_slab_runs"] is True and single["value"] > 0
    assert line["speedup"] == pytest.approx(line["value"] / single["value"], rel=1e-3)
    assert "config5" not in line          # only the 512^3 sweep takes the 1024^3 leg along
