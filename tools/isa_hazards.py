#!/usr/bin/env python3
"""Static check of the hand-scheduled solver kernels (cuda-flow3d_amd/csrc/f3d_solve.hip).

Their loads are inline assembly and their waits are counted by hand, so the compiler does not know that a load's
destination register is not valid yet: a register copy it inserts between a load and the wait that covers it (for a tied
asm operand, at a control-flow merge, ...) reads whatever the register held before -- right most of the time, wrong when the
memory system is slow.  This script compiles the file to assembly and walks every kernel in program order:

  * `global_load_dword vN, ...` (always hand-issued) puts vN in flight;
  * `s_waitcnt vmcnt(0)` lands everything;
  * the source hands registers back right behind a wait with a statement that leaves the comment `; f3d_handback vA vB ...`
    in the code: those registers have landed;
  * an instruction that names a register in flight is accepted only between an `s_waitcnt vmcnt(K)` and the end of its basic
    block with no memory instruction in between (the compiler's copies for the hand-back operands); the register then
    counts as landed.  Anywhere else it is a violation.
The walk follows the assembler's layout, not the control-flow graph: a union over CFG paths drowns in infeasible ones (the
two arms of `if (edge) wait(20) else wait(11)` are correlated with later branches), so a cold block that the compiler
places out of line can raise a false alarm -- rebuild with -gline-tables-only and read the inlined-at chain of the two
instructions before believing it (that is how the one case seen so far, a variant of k_sweep6, was cleared).
The text of a kernel is walked twice so that loads issued at the bottom of the unrolled loop meet the wait at its top.
Also checks that no kernel uses scratch (a spill of an in-flight register would be the same bug).
Usage: isa_hazards.py [file.s]   (without an argument the source is compiled with the product's flags)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "cuda-flow3d_amd", "csrc", "f3d_solve.hip")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-I" + os.path.join(ROOT, "include"),
         "-I" + os.path.join(ROOT, "cuda-flow3d_amd", "csrc"), "-S", "--cuda-device-only"]
KERNELS = ("k_sweep6", "k_phiksi6", "k_sweep7")


def compile_to_asm():
    out = os.path.join(tempfile.mkdtemp(prefix="f3d_isa_"), "f3d_solve.s")
    subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + FLAGS + [SRC, "-o", out], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return out


def vregs(text):
    regs = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", text):
        regs.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"(?<![\w\[])v(\d+)\b", text):
        regs.add(int(m.group(1)))
    return regs


def kernels(path):
    name, body = None, []
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m and any(k in m.group(1) for k in KERNELS):
            name, body = m.group(1), []
            continue
        if name:
            body.append(line.rstrip("\n"))
            if "s_endpgm" in line:
                yield name, body
                name = None


def check_kernel(body):
    flight = {}            # register -> line of the load
    after_wait = False
    bad = []
    for rep in range(2):
        for ln, raw in enumerate(body, 1):
            t = raw.strip()
            if t.startswith("; f3d_handback"):
                if after_wait:
                    for r in vregs(t):
                        flight.pop(r, None)
                else:
                    bad.append((ln, t, sorted(vregs(t) & set(flight))))
                continue
            if not t or t[0] in ";.":
                continue
            if t.endswith(":"):
                after_wait = False
                continue
            op, _, rest = t.partition(" ")
            if op == "s_waitcnt":
                m = re.search(r"vmcnt\((\d+)\)", rest)
                if m:
                    after_wait = True
                    if int(m.group(1)) == 0:
                        flight.clear()
                continue
            if op in ("s_branch", "s_setpc_b64") or op.startswith("s_cbranch"):
                after_wait = False
                continue
            if op == "global_load_dword":
                dest, addr = rest.split(",", 1)
                hit = vregs(addr) & set(flight)
                if hit:
                    bad.append((ln, t, sorted(hit)))
                flight[int(re.search(r"v(\d+)", dest).group(1))] = ln
                after_wait = False
                continue
            if op.startswith("global_") or op.startswith("buffer_"):
                after_wait = False
            hit = vregs(rest) & set(flight)
            if not hit:
                continue
            if after_wait:
                for r in hit:
                    del flight[r]
            elif rep == 0 or ln < 400:   # second walk: only the top of the loop is new information
                bad.append((ln, t, sorted(hit), [flight[h] for h in hit]))
    return bad


def scratch_use(path):
    out = {}
    cur = None
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1)
        m = re.search(r";\s*ScratchSize:\s*(\d+)", line)
        if m and cur and any(k in cur for k in KERNELS):
            out[cur] = int(m.group(1))
    return out


def run(path=None):
    path = path or compile_to_asm()
    report = {}
    for name, body in kernels(path):
        if re.search(r"ILi\d+ELi[1-9]", name) and "k_sweep7" in name:
            continue   # ablation / probe instantiations: timing experiments, not shipped results
        if "k_sweep6ILi" in name and "ILi0E" not in name:
            continue
        report[name] = check_kernel(body)
    return report, scratch_use(path)


if __name__ == "__main__":
    rep, scratch = run(sys.argv[1] if len(sys.argv) > 1 else None)
    rc = 0
    for name, bad in rep.items():
        print(f"{name}: {len(bad)} violation(s)")
        for b in bad[:20]:
            print("   ", b)
            rc = 1
    for name, size in scratch.items():
        if size:
            print(f"{name}: uses {size} bytes of scratch per lane")
            rc = 1
    sys.exit(rc)
