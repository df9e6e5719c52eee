#!/usr/bin/env python3
"""Which library do the anonymous frames of a native stack trace belong to, when no load map was kept?

Load addresses are randomised but page aligned, so the low 12 bits of a return address are the same in every run, and the
DISTANCES between frames inside one library are exact.  For a run of consecutive frames f_0 .. f_n assumed to lie in one
library, every f_i must be a return site (the address behind a call instruction) of that library at one common page-aligned
base.  This script disassembles candidate libraries, collects their return sites and looks for such a base; with a dozen
frames a match is unique for practical purposes.  Used for profiles/r02_pmc_only_crash_stack.txt.

  python3 tools/resolve_frames.py --frames 0x7b5e0da186a7,0x7b5e0da18d19,... --libs /opt/rocm/lib/libamdhip64.so ...
"""
import argparse
import bisect
import re
import subprocess

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def return_sites(lib):
    """offsets of the instructions that follow a call, and a sorted list of (function start, name)"""
    proc = subprocess.Popen([OBJDUMP, "-d", "--no-show-raw-insn", "-C", lib], stdout=subprocess.PIPE, text=True, errors="replace")
    sites, funcs = set(), []
    after_call = False
    pat = re.compile(r"^\s*([0-9a-f]+):\s+(\S+)")
    fpat = re.compile(r"^([0-9a-f]+) <(.*)>:$")
    for line in proc.stdout:
        m = pat.match(line)
        if not m:
            f = fpat.match(line)
            if f:
                funcs.append((int(f.group(1), 16), f.group(2)))
            continue
        addr = int(m.group(1), 16)
        if after_call:
            sites.add(addr)
        after_call = m.group(2).startswith("call")
    proc.wait()
    funcs.sort()
    return sites, funcs


def name_of(funcs, off):
    i = bisect.bisect_right(funcs, (off, "￿")) - 1
    if i < 0:
        return "?"
    return f"{funcs[i][1]}+0x{off - funcs[i][0]:x}"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", required=True, help="comma-separated return addresses assumed to share a library")
    ap.add_argument("--libs", nargs="+", required=True)
    a = ap.parse_args()
    frames = [int(x, 16) for x in a.frames.split(",")]
    f0 = frames[0]
    for lib in a.libs:
        sites, funcs = return_sites(lib)
        hits = []
        for r in sites:
            if (r & 0xfff) != (f0 & 0xfff):
                continue
            base = f0 - r
            if all((f - base) in sites for f in frames[1:]):
                hits.append(base)
        print(f"{lib}: {len(sites)} return sites, {len(hits)} base(s) put all {len(frames)} frames on return sites")
        for base in hits[:3]:
            print(f"  load base 0x{base:x}")
            for f in frames:
                print(f"    0x{f:x} -> +0x{f - base:x}  {name_of(funcs, f - base)}")


if __name__ == "__main__":
    main()
