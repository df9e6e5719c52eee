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

k_pair8 (csrc/f3d_solve_pair8.h; the launches that make ~90 % of a solve) has no load with a register destination: a LOADER
wave feeds an LDS ring with `global_load_lds_dwordx4` (the destination is M0 + 16 x lane), keeps one plane in flight behind a
counted `s_waitcnt vmcnt(kPerPlane)`, and the compute waves read the ring with ds_read after the workgroup barrier that follows
that wait.  check_pair8() walks the control-flow graph of every shipped instantiation (TY 4 / 8 / 12 x two sweeps / sweep +
phi/ksi x frames / frame derivatives; the ablation builds are timing experiments) under the one assumption that EXEC is not zero
where a DMA instruction sits (every piece of a plane has at least one live lane by construction: 64 h < kHaloLanes), and demands
  P1  no scratch;
  P2  every global_load_lds_* is directly preceded by its `s_nop` (>= 4 wait states with an SGPR base -- a VALU write of the
      base pair by v_readlane may sit in front of it --, >= 0 with `off`: the M0 write in front of the statement needs one);
  P3  every counted wait `s_waitcnt vmcnt(N > 0)` has N = kPerPlane of the instantiation (NA * ceil((TY + 4) / 4) row pieces +
      ceil(NA * 2 * (TY + 4) / 64) halo pieces, NA = 10 or 12) <= 63 (vmcnt is a 6-bit counter), and on every path from the closest
      workgroup barrier to the wait exactly N or 2 N DMA instructions are issued (one plane; two in the first step of a chunk)
      and no other vector-memory instruction (it would be counted by vmcnt too); the frame-derivative builds issue the pieces of a
      centre-only plane in front of each ring plane, and the count includes them;
  P4  no DMA instruction reaches a barrier or the end of the program without a vmcnt wait behind it (a plane published to the
      compute waves before it has landed);
  P5  no ds_read is reachable from the kernel entry without crossing a workgroup barrier (the prologue planes);
  P6  the hand-issued global_store_dword of the compute waves are followed by `s_waitcnt vmcnt(0)` before s_endpgm.
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


# ---- k_pair8 -------------------------------------------------------------------------------------------------------------

PAIR8 = "k_pair8"
VMEM_PREFIXES = ("global_", "buffer_", "flat_", "scratch_")


def pair8_params(name):
    """(MODE, TY, ABL, FD, YM) from the mangled name ..k_pair8ILi<MODE>ELi<TY>ELi<ABL>ELb<FD>ELb<YM>EE.."""
    m = re.search(r"k_pair8ILi(\d+)ELi(\d+)ELi(\d+)ELb([01])ELb([01])E", name)
    return tuple(int(g) for g in m.groups()) if m else None


def pair8t_params(name):
    """(MODE, TY) of a k_pair8t instantiation (thin volumes marched along y, tile without halo rows): ..k_pair8tILi<MODE>ELi<TY>EE.."""
    m = re.search(r"k_pair8tILi(\d+)ELi(\d+)E", name)
    return tuple(int(g) for g in m.groups()) if m else None


def pair8t_per_plane(ty):
    """ten inputs, TY ring rows in pieces of four, x-halo pieces of the TY rows on both sides"""
    return 10 * ((ty + 3) // 4) + (10 * 2 * ty + 63) // 64


def pair8_per_plane(ty, fd):
    """DMA instructions per plane of the three-slot ring: the count the steady-state wait must carry.  The frame-derivative builds
    keep only their seven stencilled inputs there"""
    na = 7 if fd else 10
    nj = ty + 4
    return na * ((nj + 3) // 4) + (na * 2 * nj + 63) // 64


def pair8_centre_per_plane(ty, fd):
    """frame-derivative builds: DMA instructions per plane of the two-slot ring of the five centre-only inputs (rows y0-1 .. y0+TY and
    the halo columns of the core rows); they are issued in front of the ring's plane, so they add to what a barrier-to-wait path issues"""
    if not fd:
        return 0
    return 5 * ((ty + 2 + 3) // 4) + (5 * 2 * ty + 63) // 64


# ---- k_tri: the same loader idiom (csrc/f3d_solve_tri.h), no halo pieces, no centre-only ring --------------------------------

TRI = "k_tri"


def tri_params(name):
    """(MODE, TY) from the mangled name ..k_triILi<MODE>ELi<TY>EE.."""
    m = re.search(r"k_triILi(\d+)ELi(\d+)E", name)
    return tuple(int(g) for g in m.groups()) if m else None


def tri_per_plane(ty):
    """DMA instructions per plane: ten inputs, ring rows y0-3 .. y0+TY+2 in pieces of four rows"""
    return 10 * ((ty + 6 + 3) // 4)


def instructions(body):
    """[(line number, label or None, opcode, operands)] of a kernel body; labels attach to the next instruction"""
    out, pending = [], []
    for ln, raw in enumerate(body, 1):
        t = raw.split(";")[0].strip() if not raw.strip().startswith(";") else ""
        if not t or t[0] == ".":
            m = re.match(r"^(\.LBB\w+):", t)
            if m:
                pending.append(m.group(1))
            continue
        m = re.match(r"^(\.?\w+):$", t)
        if m:
            pending.append(m.group(1))
            continue
        op, _, rest = t.partition(" ")
        out.append((ln, tuple(pending), op, rest.strip()))
        pending = []
    return out


def is_dma(op):
    return op.startswith("global_load_lds_")


def build_cfg(ins, exec_nonzero):
    """successor lists by instruction index.  exec_nonzero: a `s_cbranch_execz` that jumps over a DMA instruction is never taken
    and a `s_cbranch_execnz` into a block that holds one always is (EXEC != 0 wherever a piece of a plane is issued)"""
    label_at = {}
    for i, (_, labels, _, _) in enumerate(ins):
        for l in labels:
            label_at[l] = i

    def block_has_dma(i):   # straight-line run from instruction i to the next branch
        while i < len(ins):
            op = ins[i][2]
            if is_dma(op):
                return True
            if op == "s_branch" or op.startswith("s_cbranch") or op == "s_endpgm":
                return False
            i += 1
        return False

    succ = [[] for _ in ins]
    for i, (_, _, op, rest) in enumerate(ins):
        nxt = [i + 1] if i + 1 < len(ins) else []
        if op == "s_endpgm":
            continue
        if op == "s_branch":
            succ[i] = [label_at[rest]] if rest in label_at else []
        elif op.startswith("s_cbranch"):
            tgt = [label_at[rest]] if rest in label_at else []
            if exec_nonzero and op == "s_cbranch_execz" and nxt and block_has_dma(nxt[0]):
                succ[i] = nxt
            elif exec_nonzero and op == "s_cbranch_execnz" and tgt and block_has_dma(tgt[0]):
                succ[i] = tgt
            else:
                succ[i] = nxt + tgt
        else:
            succ[i] = nxt
    return succ


def vmcnt_of(op, rest):
    if op != "s_waitcnt":
        return None
    m = re.search(r"vmcnt\((\d+)\)", rest)
    return int(m.group(1)) if m else None


def check_pair8(body, per_plane, centre=0):
    """violations of P2 .. P6 in one k_pair8 body (list of assembly lines); per_plane = the wait count the source must carry,
    centre = DMA instructions of a centre-only plane issued in front of every ring plane (frame-derivative builds)"""
    ins = instructions(body)
    bad = []
    n = len(ins)
    # P2
    for i, (ln, _, op, rest) in enumerate(ins):
        if not is_dma(op):
            continue
        pln, _, pop, prest = ins[i - 1] if i else (0, (), "", "")
        need = 0 if rest.endswith("off") else 4
        if pop != "s_nop" or int(prest or -1) < need:
            bad.append(("P2", ln, f"{op} {rest}: wants `s_nop {need}` (or more) directly in front, found `{pop} {prest}`"))
    succ = build_cfg(ins, exec_nonzero=True)
    # P3 + P4: forward from the kernel entry and from every workgroup barrier, counting the DMA instructions issued on the way.
    # The walk is path sensitive in one respect: the compiler materialises `continue` / `break` decisions of the loader loop as
    # s_mov_b64 s[a:b], -1 | 0 ... s_and_b64 vcc, exec, s[a:b]; s_cbranch_vccnz -- with EXEC != 0 such a branch has one feasible
    # arm, and following the other one pairs an issue with the wrong wait.
    def sregs(text):
        regs = set()
        for m in re.finditer(r"\bs\[(\d+):(\d+)\]", text):
            regs.update(range(int(m.group(1)), int(m.group(2)) + 1))
        for m in re.finditer(r"(?<![\w\[])s(\d+)\b", text):
            regs.add(int(m.group(1)))
        return regs

    reported = set()

    def report(rule, ln, text):
        if (rule, ln) not in reported:
            reported.add((rule, ln))
            bad.append((rule, ln, text))

    seeds = [0] + [j for i, (_, _, op, _) in enumerate(ins) if op == "s_barrier" for j in succ[i]]
    seen = set()
    waits_reached = set()
    stack = [(i, 0, False, frozenset(), None) for i in seeds]   # instruction, DMA count, other VMEM seen, known pairs, vcc
    while stack:
        state = stack.pop()
        if state in seen:
            continue
        seen.add(state)
        i, count, other, known, vcc = state
        ln, _, op, rest = ins[i]
        if op in ("s_barrier", "s_endpgm"):
            if count:
                report("P4", ln, f"`{op}` is reached with {count} DMA instruction(s) issued and no vmcnt wait behind them")
            continue
        cnt = vmcnt_of(op, rest)
        if cnt is not None:
            if cnt > 0:
                waits_reached.add(i)
                if cnt != per_plane or cnt > 63:
                    report("P3", ln, f"s_waitcnt vmcnt({cnt}): the instantiation issues {per_plane} pieces per plane (and vmcnt holds 63)")
                elif count not in (cnt + centre, 2 * cnt + centre):
                    report("P3", ln, f"s_waitcnt vmcnt({cnt}) is reached with {count} DMA instructions issued since the barrier "
                                     f"(wants {cnt + centre} or {2 * cnt + centre}: one plane stays in flight)")
                elif other:
                    report("P3", ln, f"another vector-memory instruction sits between the barrier and s_waitcnt vmcnt({cnt}): vmcnt counts it too")
            count, other = 0, False
        elif is_dma(op):
            count += 1
        elif op.startswith(VMEM_PREFIXES):
            other = True
        # the little that is tracked about scalar registers
        m = re.match(r"s_mov_b64 s\[(\d+):(\d+)\], (-1|0)$", f"{op} {rest}")
        if m:
            pair = (int(m.group(1)), int(m.group(2)))
            known = frozenset([k for k in known if k[0] != pair] + [(pair, int(m.group(3)))])
        else:
            m = re.match(r"s_and_b64 vcc, exec, s\[(\d+):(\d+)\]$", f"{op} {rest}")
            if m:
                pair = (int(m.group(1)), int(m.group(2)))
                val = dict(known).get(pair)
                vcc = None if val is None else (val != 0)
            elif not op.startswith(("s_cbranch", "s_branch")):
                touched = sregs(rest)
                if touched:
                    known = frozenset(k for k in known if not (set(range(k[0][0], k[0][1] + 1)) & touched))
                if "vcc" in rest or op.startswith("v_cmp") or op.startswith("v_div_scale") or op.startswith("v_add_co") or \
                        op.startswith("v_sub_co") or op.startswith("v_addc") or op.startswith("v_subb"):
                    vcc = None
        nxt = succ[i]
        if op in ("s_cbranch_vccnz", "s_cbranch_vccz") and vcc is not None and len(nxt) == 2:
            taken = (op == "s_cbranch_vccnz") == vcc
            nxt = [nxt[1]] if taken else [nxt[0]]     # build_cfg lists the fall-through first
        for j in nxt:
            stack.append((j, count, other, known, vcc))
    counted = [i for i, (_, _, op, rest) in enumerate(ins) if (vmcnt_of(op, rest) or 0) > 0]
    if any(is_dma(op) for _, _, op, _ in ins) and not counted:
        report("P3", 0, "the kernel issues DMA instructions but holds no counted vmcnt wait: not the loader this check was written for")
    for i in counted:
        if i not in waits_reached:
            report("P3", ins[i][0], "a counted vmcnt wait the walk never reached: the control-flow assumptions of the check no longer hold")
    # P5 / P6 on the full graph (no assumption about EXEC)
    full = build_cfg(ins, exec_nonzero=False)
    seen, stack = set(), [0] if ins else []
    while stack:
        i = stack.pop()
        if i in seen:
            continue
        seen.add(i)
        _, _, iop, irest = ins[i]
        if iop == "s_barrier":
            continue
        if iop.startswith("ds_read"):
            bad.append(("P5", ins[i][0], f"`{iop} {irest}` can run before the first workgroup barrier"))
            break
        stack.extend(full[i])
    for st, (ln, _, op, rest) in enumerate(ins):
        if not op.startswith("global_store"):
            continue
        seen, stack = set(), list(full[st])
        while stack:
            i = stack.pop()
            if i in seen:
                continue
            seen.add(i)
            _, _, iop, irest = ins[i]
            if vmcnt_of(iop, irest) == 0:
                continue
            if iop == "s_endpgm":
                bad.append(("P6", ln, f"{op} at line {ln} reaches s_endpgm without s_waitcnt vmcnt(0)"))
                stack = []
                break
            stack.extend(full[i])
        if bad and bad[-1][0] == "P6":
            break
    return bad


def pair8_kernels(path, key=PAIR8):
    name, body = None, []
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m and key in m.group(1):
            name, body = m.group(1), []
            continue
        if name:
            body.append(line.rstrip("\n"))
            if re.match(r"^\.Lfunc_end", line):
                yield name, body
                name = None


def run_pair8(path=None):
    """{kernel: violations} for every shipped k_pair8 instantiation, and {kernel: scratch bytes}"""
    path = path or compile_to_asm()
    report, scratch = {}, {}
    cur = None
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1)
        m = re.search(r";\s*ScratchSize:\s*(\d+)", line)
        if m and cur and PAIR8 in cur:
            scratch[cur] = int(m.group(1))
    for name, body in pair8_kernels(path):
        tight = pair8t_params(name)
        if tight is not None:
            report[name] = check_pair8(body, pair8t_per_plane(tight[1]))
            continue
        prm = pair8_params(name)
        if prm is None or prm[2] != 0:
            continue        # ablation builds (ABL != 0): timing experiments with wrong results, never launched by default
        report[name] = check_pair8(body, pair8_per_plane(prm[1], prm[3]), pair8_centre_per_plane(prm[1], prm[3]))
    return report, {k: v for k, v in scratch.items() if pair8t_params(k) is not None or (pair8_params(k) or (0, 0, 1, 0))[2] == 0}


def run_tri(path=None):
    """{kernel: violations} for every k_tri instantiation (all of them ship), and {kernel: scratch bytes}"""
    path = path or compile_to_asm()
    report, scratch = {}, {}
    cur = None
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1)
        m = re.search(r";\s*ScratchSize:\s*(\d+)", line)
        if m and cur and TRI in cur:
            scratch[cur] = int(m.group(1))
    for name, body in pair8_kernels(path, TRI):
        prm = tri_params(name)
        if prm is not None:
            report[name] = check_pair8(body, tri_per_plane(prm[1]))
    return report, scratch


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
    asm = sys.argv[1] if len(sys.argv) > 1 else compile_to_asm()
    rep, scratch = run(asm)
    rep8, scratch8 = run_pair8(asm)
    rep.update(rep8)
    scratch.update(scratch8)
    rep3, scratch3 = run_tri(asm)
    rep.update(rep3)
    scratch.update(scratch3)
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
