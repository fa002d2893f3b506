#!/usr/bin/env python3
"""lk.hip keeps the derivative tile's loads in flight IN REGISTERS while the template patch is computed
(tile_issue / dtile_issue ... tile_commit<.., N> / dtile_commit).  The compiler does not know that those
registers are being written asynchronously: if it ever copies or reuses one of them between the load and the
s_waitcnt that covers it, the kernel reads or clobbers data in flight and nothing but the results would tell.
This check reads the gfx950 assembly of lk.hip and fails if, for any hand-issued global_load_dwordx4, an
instruction between the load and the first following `s_waitcnt vmcnt(N)` that is guaranteed to cover it touches
the load's destination registers.

    python tools/check_lk_inflight.py --so libsvo_hip.so   the SHIPPED library: its gfx950 code objects are
                                                           extracted and disassembled (llvm-objdump); this is what
                                                           csrc/Makefile runs after linking, and a violation fails
                                                           the build
    python tools/check_lk_inflight.py lk.s                 an assembly listing (hipcc -S or llvm-objdump -d)
Checked: every lk_track_kernel<C> of the listing (the only kernels with hand-issued loads); none found = failure.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ros_stereo_slam_amd", "csrc")
OBJDUMP = os.environ.get("LLVM_OBJDUMP", "/opt/rocm/lib/llvm/bin/llvm-objdump")


def disassemble_so(so_path):
    """gfx950 disassembly of every code object bundled in the library, as one listing."""
    import shutil

    tmp = tempfile.mkdtemp(prefix="lkso")
    try:
        local = os.path.join(tmp, "lib.so")
        shutil.copy(so_path, local)
        subprocess.run([OBJDUMP, "--offloading", local], check=True, stdout=subprocess.DEVNULL, cwd=tmp)
        parts = sorted(f for f in os.listdir(tmp) if "amdgcn" in f)
        if not parts:
            raise SystemExit(f"{so_path}: no gfx950 code object found")
        text = []
        for f in parts:
            r = subprocess.run([OBJDUMP, "-d", "--symbolize-operands", os.path.join(tmp, f)], check=True,
                               capture_output=True, text=True)
            text.append(r.stdout)
        out = os.path.join(tempfile.mkdtemp(prefix="lkasm"), "libsvo_hip.dis")
        with open(out, "w") as fh:
            fh.write("\n".join(text))
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def regs_of(text):
    """VGPR numbers an operand string mentions (v12, v[12:15])."""
    s = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", text):
        s.update(range(int(a), int(b) + 1))
    s.update(int(a) for a in re.findall(r"\bv(\d+)\b", text))
    return s


LABEL = re.compile(r"^(?:[0-9a-f]+ )?(\.LBB\S*|<L\d+>):")
KERNEL = re.compile(r"^(?:[0-9a-f]+ <)?(_ZN\S*lk_track_kernel[^>:\s]*)>?:")
SYMBOL = re.compile(r"^(?:[0-9a-f]+ <[^>]+>|[A-Za-z_$][\w$.]*):")


def check(path):
    """Only lk.hip issues loads by hand (asm volatile), and only in lk_track_kernel<C>: there EVERY
    global_load_dwordx4 is hand-issued, so the kernel's own vmcnt bookkeeping can be replayed exactly.  Kernels the
    compiler schedules by itself are not checked (it knows about its own loads)."""
    lines = [l.split("//")[0].rstrip() for l in open(path).read().split("\n")]   # objdump puts the encoding there
    problems, kernels, loads_seen = [], 0, 0
    i = 0
    while i < len(lines):
        m = KERNEL.match(lines[i])
        if not m:
            i += 1
            continue
        kernels += 1
        name = m.group(1)
        # the kernel ends where the next symbol begins (an early exit puts an s_endpgm in the middle of it)
        end = next((j for j in range(i + 1, len(lines)) if SYMBOL.match(lines[j]) and not LABEL.match(lines[j])), len(lines))
        body = [(j, lines[j].strip()) for j in range(i + 1, end)]
        body = [(j, l) for j, l in body if l and not l.startswith(";") and not l.startswith(".") or LABEL.match(l)]
        # outstanding hand-issued loads, oldest first: (line, dest registers)
        pending = []
        for j, l in body:
            if LABEL.match(l):
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
            if op.startswith("s_cbranch") or op == "s_branch" or op == "s_endpgm":
                if pending:
                    problems.append(f"{name}: branch at line {j + 1} inside an in-flight window (loads from line {pending[0][0] + 1})")
                    pending = []
                continue
            if pending:
                touched = regs_of(l)
                for lj, d in pending:
                    if touched & d:
                        problems.append(f"{name}: line {j + 1} touches v{sorted(touched & d)} in flight since line {lj + 1}: {l}")
        i = end  # the next symbol's own line (it may be the next tracking kernel)
    return kernels, loads_seen, problems


def main():
    if len(sys.argv) == 3 and sys.argv[1] == "--so":
        path = disassemble_so(sys.argv[2])
    elif len(sys.argv) == 2:
        path = sys.argv[1]
    else:
        print(__doc__)
        return 2
    kernels, loads, problems = check(path)
    for p in problems:
        print(p)
    print(f"{path}: {kernels} tracking kernel(s), {loads} hand-issued loads, {len(problems)} problem(s)")
    return 1 if problems or kernels == 0 or loads == 0 else 0


if __name__ == "__main__":
    sys.exit(main())
