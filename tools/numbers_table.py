#!/usr/bin/env python3
"""The one table of current numbers in DESIGN.md section 7, generated from the committed records under profiles/ (round tag below).
   python tools/numbers_table.py            prints the table
   python tools/numbers_table.py --write    replaces the block between the two `numbers` markers in DESIGN.md
Every row names the file it was read from; a row whose file is missing is left out (nothing is typed in by hand)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = "r04"
P = lambda name: os.path.join(ROOT, "profiles", f"{TAG}_{name}")


def last_json_line(path):
    with open(path) as f:
        lines = [l for l in f.read().splitlines() if l.startswith("{")]
    return json.loads(lines[-1])


def rows():
    out = []
    f = P("bench_steps20_warmup5.json")
    if os.path.exists(f):
        b = last_json_line(f)
        r = b["roofline"]
        src = os.path.relpath(f, ROOT)
        out.append(("512³ full default pyramid, 1 GPU (`bench.py --gpus 1 --steps 20 --warmup 5`)",
                    f"**{b['value']:.2f} Mvoxels/s**, {b['ms_per_step']:.1f} ms per solve device-resident; {b['host_inclusive']['value']:.2f} with "
                    f"H2D + D2H inside the timer; whole run {100 * b['whole_run_roofline_frac']:.1f} % of 8 TB/s in algorithmic bytes; "
                    f"`parity.match` {b['parity']['match']}", src))
        fin = r["finest_level"]
        out.append(("two-sweep launch `k_pair8<SS>` (dominant kernel)",
                    f"over the pyramid {r['achieved']:.0f} GB/s algorithmic = **{r['frac']:.3f}** ({r['avg_launch_us']:.1f} µs average over "
                    f"{r['launches']} launches); 512³ level {fin['avg_launch_us']:.0f} µs = {fin['frac']:.3f}; in counter bytes "
                    f"{r.get('traffic', 0) / 1e9:.2f} GB per 512³ launch → {r.get('hbm_GBs', 0):.0f} GB/s = `hbm_frac` {r.get('hbm_frac', 0):.3f}", src))
        sp = r["sweep_phi_ksi"]
        out.append(("sweep + next phi/ksi `k_pair8<SP>`",
                    f"over the pyramid {sp['achieved']:.0f} GB/s algorithmic = {sp['achieved'] / 8000:.3f}; 512³ level "
                    f"{sp['finest_level']['avg_launch_us']:.0f} µs = {sp['finest_level']['frac']:.3f}", src))
        out.append(("all solver launches at their algorithmic bytes", f"{r['all_solver_launches']['achieved']:.0f} GB/s = "
                    f"{r['all_solver_launches']['frac']:.3f}", src))
        for c in b.get("configs", []):
            out.append((c["workload"], f"{c['ms_per_step']:.1f} ms = {c['value']:.1f} Mvoxels/s = {c['roofline_frac']:.3f} of the roofline", src))
        if "fixed_sample" in b:
            fs = b["fixed_sample"]
            cpu = fs.get("cpu", {})
            out.append(("SURVEY §8d fixed sample (phi/ksi + 5 sweeps on the 512³ level)",
                        f"GPU {fs['gpu']['value'] / 1e3:.1f} Gvoxel-updates/s ({fs['gpu']['ms']:.2f} ms); host cores "
                        f"{cpu.get('value', 0) / 1e3:.2f} ({cpu.get('cores', '?')} threads)", src))
        if "cpu_baseline" in b:
            cb = b["cpu_baseline"]
            out.append(("`cpu_baseline` (BASELINE config 2 in full on the host, oracle)", f"{cb['value']:.4f} Mvoxels/s on {cb['cores']} threads "
                        f"(`kind: \"{cb['kind']}\"`)", src))
    f = P("pmc_traffic.json")
    if os.path.exists(f):
        d = json.load(open(f))
        cells = []
        for k in ("k_pair8_fd", "k_pair8_sweep_phi_ksi_fd", "k_pair8", "k_pair8_sweep_phi_ksi", "k_sweep6", "k_phiksi6"):
            if k in d:
                cells.append(f"`{k}` {d[k]['hbm_bytes_per_launch'] / 1e9:.2f} GB ({d[k]['ratio']:.2f} × algorithmic)")
        out.append(("HBM bytes per 512³ launch (PMC: 2 × FETCH_SIZE + WRITE_SIZE, separate passes)", "; ".join(cells), os.path.relpath(f, ROOT)))
    f = P("c5_one_gpu.json")
    if os.path.exists(f):
        b = last_json_line(f)
        out.append(("1024³ (BASELINE config 5) on ONE GPU", f"{b['value']:.2f} Mvoxels/s, {b['ms_per_step'] / 1e3:.2f} s per solve; `parity.match` "
                    f"{b['parity']['match']}", os.path.relpath(f, ROOT)))
    f = P("bench_gpus2_bare_shm_rehearsal.json")
    if os.path.exists(f):
        b = last_json_line(f)
        o = b["exchange_orders"]
        cell = (f"REHEARSAL of the line's orchestration, two rank processes on one GPU over shared memory (not a number): both orders "
                f"`parity.match` {[o[k]['parity']['match'] for k in o]}, `single_gpu_same_size.digest_equals_the_slab_runs` "
                f"{b['single_gpu_same_size']['digest_equals_the_slab_runs']}, `speedup` field present ({b['speedup']}), exchanges per step "
                f"{[o[k]['comm']['exchanges_per_step_rank0'] for k in o]}, `config5` leg "
                f"{'present, parity ' + str(b['config5']['parity']['match']) if 'config5' in b else 'not run'}")
        out.append(("`bench.py --gpus 2` started bare", cell, os.path.relpath(f, ROOT)))
    f = P("piecemeal_1024_16gb.txt")
    if os.path.exists(f):
        import re
        secs = re.findall(r"piecemeal:\s+([\d.]+) s", open(f).read())
        if len(secs) >= 2:
            out.append(("out-of-core `OpticalFlowP`, 1024³ on a 16 GB budget (`tools/pbench.py`; same bits as the resident driver)",
                        f"{float(secs[1]):.2f} s with the flow update inside the solver's last residency, {float(secs[0]):.2f} s with the separate "
                        f"add operator (round 3: 46.8 s)", os.path.relpath(f, ROOT)))
    f = P("piecemeal_1024_16gb_registration.txt")
    if os.path.exists(f):
        import re
        secs = re.findall(r"piecemeal:\s+([\d.]+) s", open(f).read())
        if len(secs) >= 2:
            out.append(("... and with frame 1 registered inside the solver's first residency as well",
                        f"**{float(secs[1]):.2f} s**; with the separate registration operator {float(secs[0]):.2f} s (same call)",
                        os.path.relpath(f, ROOT)))
    f = P("piecemeal_huge_pages.txt")
    if os.path.exists(f):
        import re
        secs = re.findall(r"piecemeal:\s+([\d.]+) s", open(f).read())
        if len(secs) >= 2:
            out.append(("out-of-core `OpticalFlowP`, 1024³ on a 16 GB budget, FINAL state of round 4 (round 3: 46.8 s; same bits as the resident driver)",
                        f"**{float(secs[0]):.2f} s** ({float(secs[1]):.2f} s with transparent huge pages for the host scratch -- taken out again, LABBOOK)",
                        os.path.relpath(f, ROOT)))
    f = P("piecemeal_1024_16gb_handover.txt")
    if os.path.exists(f):
        import re
        secs = re.findall(r"piecemeal:\s+([\d.]+) s", open(f).read())
        if len(secs) >= 2:
            out.append(("... and the planes neighbouring chunks share handed on from chunk set to chunk set on "
                        "the device (every plane of every field over the link once per pass)",
                        f"**{float(secs[-1]):.2f} s**; with whole windows uploaded {float(secs[0]):.2f} s (same call; `results identical` to the "
                        f"resident driver; per-level solver seconds in the record)", os.path.relpath(f, ROOT)))
    f = P("piecemeal_1024_16gb_shared_buffers.txt")
    if os.path.exists(f):
        import re
        secs = re.findall(r"piecemeal:\s+([\d.]+) s", open(f).read())
        if secs:
            out.append(("... with the resample operator on two buffer sets, the host scratch prepared beside the resident levels and the solver's "
                        "two chunk sets sharing the compute-only fields",
                        f"**{float(secs[-1]):.2f} s** (`results identical` to the resident driver; per-level solver seconds in the record)",
                        os.path.relpath(f, ROOT)))
    f = P("thin_tile_solves.txt")
    if os.path.exists(f):
        import re
        ms = re.findall(r"([\d.]+) ms per solve", open(f).read())
        if len(ms) >= 2:
            out.append(("BASELINE config 3 with its thin levels on the y-marching tile without halo rows (`k_pair8t`) / with halo rows",
                        f"{float(ms[0]):.1f} ms / {float(ms[1]):.1f} ms (same call, `tools/trace_size.py --config c3`)", os.path.relpath(f, ROOT)))
    for size in (512, 1024):
        f = P(f"slab8_onegpu_{size}.json")
        if os.path.exists(f):
            d = json.load(open(f))
            tot = d.get("total", d)
            model = ""
            try:
                import subprocess
                txt = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "scale_model.py"), "--size", str(size)], capture_output=True,
                                     text=True).stdout
                ups = [l.split("speed-up")[1].strip() for l in txt.splitlines() if "speed-up" in l]
                model = f"; MODEL (`tools/scale_model.py`, 50 µs per exchange assumed): {ups[1]} / {ups[2]} / {ups[3]} × at 2 / 4 / 8 GPUs"
            except Exception:
                pass
            out.append((f"8 z-slabs of {size}³ run one after the other on ONE GPU (the work 8 GPUs divide)",
                        f"kernel time {tot.get('ratio', 0):.3f} × the unsplit solve{model}", os.path.relpath(f, ROOT)))
    return out


def table():
    lines = ["<!-- numbers:begin (generated by tools/numbers_table.py from profiles/; do not edit by hand) -->",
             "| what | measured (one MI355X; boxes of the pool differ by ±3 %: the bench line and the kernel statistics of this table are of ONE box, `tools/jobs/r4_job36.sh`; another box gave 71.33 Mvoxels/s and 0.927 for the same kernels, `profiles/r04_bench_steps20_warmup5_other_box.json`) | record |", "|---|---|---|"]
    for what, value, src in rows():
        lines.append(f"| {what} | {value} | `{src}` |")
    lines.append("<!-- numbers:end -->")
    return "\n".join(lines)


if __name__ == "__main__":
    t = table()
    if "--write" in sys.argv:
        path = os.path.join(ROOT, "DESIGN.md")
        s = open(path).read()
        if "NUMBERS_TABLE" in s:
            s = s.replace("NUMBERS_TABLE", t)
        else:
            a, b = s.index("<!-- numbers:begin"), s.index("<!-- numbers:end -->") + len("<!-- numbers:end -->")
            s = s[:a] + t + s[b:]
        open(path, "w").write(s)
    else:
        print(t)
