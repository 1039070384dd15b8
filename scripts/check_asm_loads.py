"""Build-time check of the hand-tracked loads of the photometric linearize (photo_kernels.hip: gload16 / vm_wait_keep).

The inline-asm `global_load_dwordx4` of the staged sampler (and, r05, the `buffer_load_dwordx2` basis-row loads of the contraction phase) are invisible to the compiler's wait-count bookkeeping: the
destination registers are only valid after the next inline-asm `s_waitcnt vmcnt`.  The compiler is free to copy or spill a
register between the two statements (it believes the value exists); this script compiles the kernel file to assembly and
fails if any instruction between such a load and the following inline-asm wait touches the destination registers.
Usage: python scripts/check_asm_loads.py   (dev tool; also run by tests/test_build_checks.py)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "sage_slam_amd", "csrc", "photo_kernels.hip")


def regs_of(text):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", text):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", text):
        out.add(int(m.group(1)))
    return out


def is_tracked_load(line):
    """inline-asm loads with a VGPR destination: gload16 (global_load_dwordx4) and the basis-row loads of the contraction
    phase (buffer_load_dword / dwordx2 ... offen); the LDS-direct buffer loads of the sampler have no destination register"""
    if "global_load_dwordx4" in line:
        return True
    return bool(re.search(r"\bbuffer_load_dword(x2)?\s", line)) and " lds" not in line


def check(asm_text):
    """Walks the control-flow graph from every hand-tracked load to the inline-asm waits that cover it."""
    lines = asm_text.split("\n")
    labels = {}
    for i, ln in enumerate(lines):
        m = re.match(r"^([.\w$]+):", ln)
        if m:
            labels[m.group(1)] = i
    problems, n_loads, kernel = [], 0, "?"
    for i, ln in enumerate(lines):
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            kernel = m.group(1)
        if "#ASMSTART" not in ln:
            continue
        k = i + 1
        while k < len(lines) and "#ASMEND" not in lines[k] and not is_tracked_load(lines[k]):
            k += 1
        if k >= len(lines) or not is_tracked_load(lines[k]):
            continue
        dst = regs_of(lines[k].split(",")[0])
        n_loads += 1
        while "#ASMEND" not in lines[k]:
            k += 1
        todo, seen = [k + 1], set()
        while todo:
            j = todo.pop()
            while j < len(lines) and j not in seen:
                seen.add(j)
                raw = lines[j]
                body = raw.split(";")[0].strip()
                if "#ASMSTART" in raw and j + 1 < len(lines) and "s_waitcnt vmcnt" in lines[j + 1]:
                    break  # covered on this path
                if body.startswith("s_endpgm"):
                    problems.append((kernel, i + 2, "a path reaches the end of the kernel without a wait"))
                    break
                if body and "#ASM" not in raw and regs_of(body) & dst:
                    problems.append((kernel, j + 1, body))
                m = re.match(r"^(s_cbranch_\w+|s_branch)\s+([.\w$]+)", body)
                if m and m.group(2) in labels:
                    todo.append(labels[m.group(2)])
                    if m.group(1) == "s_branch":
                        break
                j += 1
    return n_loads, problems


def main():
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "photo.s")
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                               "-I" + os.path.dirname(SRC), "-x", "hip", "--cuda-device-only", "-S", SRC, "-o", out] +
                              [a for a in sys.argv[1:] if a.startswith("-D")],   # (variant builds: -DSAGE_...)
                              stderr=subprocess.DEVNULL)
        n, problems = check(open(out).read())
    print(f"{n} hand-tracked loads checked, {len(problems)} problems")
    for k, line, what in problems:
        print(f"  {k} line {line}: {what}")
    return 1 if problems or n == 0 else 0


if __name__ == "__main__":
    sys.exit(main())
