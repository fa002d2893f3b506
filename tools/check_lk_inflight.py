#!/usr/bin/env python3
"""lk.hip keeps the derivative tile's loads in flight IN REGISTERS while the template patch is computed
(tile_issue / dtile_issue ... tile_commit<.., N> / dtile_commit).  The compiler does not know that those
registers are being written asynchronously: if it ever copies or reuses one of them between the load and the
s_waitcnt that covers it, the kernel reads or clobbers data in flight and nothing but the results would tell.
This check reads the gfx950 assembly of lk.hip and fails if, for any hand-issued global_load_dwordx4, an
instruction between the load and the first following `s_waitcnt vmcnt(N)` that is guaranteed to cover it touches
the load's destination registers.

    python tools/check_lk_inflight.py [lk.s]     (without an argument: compiles lk.hip with the library's flags)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ros_stereo_slam_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--offload-arch=gfx950", "-S", "--cuda-device-only"]


def compile_asm():
    out = os.path.join(tempfile.mkdtemp(prefix="lkasm"), "lk.s")
    subprocess.run(["hipcc", *FLAGS, "-o", out, os.path.join(CSRC, "lk.hip")], check=True, stderr=subprocess.DEVNULL)
    return out


def regs_of(text):
    """VGPR numbers an operand string mentions (v12, v[12:15])."""
    s = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", text):
        s.update(range(int(a), int(b) + 1))
    s.update(int(a) for a in re.findall(r"\bv(\d+)\b", text))
    return s


def check(path):
    lines = open(path).read().split("\n")
    problems, kernels, loads_seen = [], 0, 0
    i = 0
    while i < len(lines):
        m = re.match(r"^(_ZN\S*lk_track_kernel\S*):", lines[i])
        if not m:
            i += 1
            continue
        kernels += 1
        name = m.group(1)
        end = next(j for j in range(i, len(lines)) if "s_endpgm" in lines[j])
        body = [(j, lines[j].strip()) for j in range(i + 1, end)]
        body = [(j, l) for j, l in body if l and not l.startswith(";") and not l.startswith(".") or re.match(r"^\.LBB", l)]
        # outstanding hand-issued loads, oldest first: (line, dest registers)
        pending = []
        for j, l in body:
            if re.match(r"^\.LBB", l):
                # a label: control flow joins here; every in-flight window of lk.hip is straight-line code
                if pending:
                    problems.append(f"{name}: label at line {j + 1} inside an in-flight window (loads from line {pending[0][0] + 1})")
                    pending = []
                continue
            op = l.split()[0]
            if op == "global_load_dwordx4":
                dest = regs_of(l.split(",")[0])
                # address operands of THIS load must not be pending destinations either
                if any(regs_of(",".join(l.split(",")[1:])) & d for _, d in pending):
                    problems.append(f"{name}: line {j + 1} uses an in-flight register as an address: {l}")
                pending.append((j, dest))
                loads_seen += 1
                continue
            if op == "s_waitcnt":
                mm = re.search(r"vmcnt\((\d+)\)", l)
                if mm:
                    keep = int(mm.group(1))
                    pending = pending[len(pending) - keep:] if keep < len(pending) else pending
                    if keep == 0:
                        pending = []
                continue
            if op.startswith("s_cbranch") or op == "s_branch":
                if pending:
                    problems.append(f"{name}: branch at line {j + 1} inside an in-flight window (loads from line {pending[0][0] + 1})")
                    pending = []
                continue
            if pending:
                touched = regs_of(l)
                for lj, d in pending:
                    if touched & d:
                        problems.append(f"{name}: line {j + 1} touches v{sorted(touched & d)} in flight since line {lj + 1}: {l}")
        i = end + 1
    return kernels, loads_seen, problems


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else compile_asm()
    kernels, loads, problems = check(path)
    for p in problems:
        print(p)
    print(f"{path}: {kernels} tracking kernel(s), {loads} hand-issued loads, {len(problems)} problem(s)")
    return 1 if problems or kernels == 0 or loads == 0 else 0


if __name__ == "__main__":
    sys.exit(main())
