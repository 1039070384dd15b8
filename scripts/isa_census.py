"""Instruction census of the photometric linearize kernel per phase.

Compiles csrc/photo_kernels.hip with -DSAGE_PHASE_MARKERS (comment lines naming the phase that follows, fenced by
scheduling barriers) to gfx950 assembly and counts, inside photo_kernel<CS,FS,true,1>, the instructions between consecutive
markers by class.  The sub-tile loop body is straight-line per 64-pixel wave slice (the channel groups and the 16 pixel
groups of the contraction are unrolled), so the static count of a phase IS its count per slice; the markers' scheduling
barriers cost the marked build a few instructions of freedom (the unmarked build's total is printed next to it).

usage: python scripts/isa_census.py [CS FS [MODE]] [-o profiles/r05_photo_isa_census.txt]
"""
import os
import re
import subprocess
import sys
import tempfile
from collections import OrderedDict, Counter

JACB = 0 if os.environ.get("CENSUS_ERROR_PASS") == "1" else 1   # CENSUS_ERROR_PASS=1: photo_kernel<CS,FS,false,MODE>
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "sage_slam_amd", "csrc", "photo_kernels.hip")


def compile_asm(markers, extra=()):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "photo.s")
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
               "-I" + os.path.dirname(SRC), "-x", "hip", "--cuda-device-only", "-S", SRC, "-o", out] + list(extra)
        if markers:
            cmd.append("-DSAGE_PHASE_MARKERS=1")
        subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
        return open(out).read()


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_pk_"):
        return "valu_pk"
    if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
        return "valu_lane"
    if op.endswith("_dpp") or "dpp" in op:
        return "valu_dpp"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith(("s_waitcnt", "s_nop", "s_setprio", "s_barrier", "s_sleep")):
        return "wait_nop"
    if op.startswith(("s_cbranch", "s_branch", "s_endpgm")):
        return "branch"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    return "other"


def kernel_body(asm, CS, FS, MODE=1):
    key = f"photo_kernelILi{CS}ELi{FS}ELb{JACB}ELi{MODE}EE"
    lines = asm.split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + key + r"\w*:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    return lines[start:end + 1], lines


def census(body):
    phases = OrderedDict()
    cur = "prologue"
    phases[cur] = Counter()
    ops = OrderedDict()
    for ln in body:
        m = re.search(r"; SAGE_PHASE (\w+)", ln)
        if m:
            cur = m.group(1)
            phases.setdefault(cur, Counter())
            continue
        t = ln.split(";")[0].strip()
        if not t or t.endswith(":") or t.startswith((".", "#")):
            continue
        op = t.split()[0]
        c = classify(op)
        phases[cur][c] += 1
        ops.setdefault(cur, Counter())[op] += 1
    return phases, ops


def vgprs(lines, CS, FS, MODE=1):
    """(VGPRs, VGPR spills, SGPR spills) from the kernel's metadata record"""
    key = f"photo_kernelILi{CS}ELi{FS}ELb{JACB}ELi{MODE}EE"
    txt = "\n".join(lines)
    i = txt.find(".name:", txt.find("amdhsa.kernels"))
    m = re.search(r"\.name:\s+_Z\w*" + key + r"\w*\n([\s\S]*?)\.wavefront_size", txt)
    if not m:
        return (-1, -1, -1)
    blk = m.group(1)
    g = lambda k: int(re.search(k + r":\s*(\d+)", blk).group(1))
    return (g(r"\.vgpr_count"), g(r"\.vgpr_spill_count"), g(r"\.sgpr_spill_count"))


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("-")]
    out_path = None
    if "-o" in sys.argv:
        out_path = sys.argv[sys.argv.index("-o") + 1]
        args = [a for a in args if a != out_path]
    CS, FS = (int(args[0]), int(args[1])) if len(args) >= 2 else (32, 16)
    MODE = int(args[2]) if len(args) >= 3 else 1
    extra = [a for a in sys.argv[1:] if a.startswith("-D")]
    body, lines = kernel_body(compile_asm(True, extra), CS, FS, MODE)
    phases, ops = census(body)
    plain_body, plain_lines = kernel_body(compile_asm(False, extra), CS, FS, MODE)
    plain, _ = census(plain_body)
    cols = ["valu", "valu_pk", "valu_dpp", "valu_lane", "mfma", "lds", "vmem", "salu", "smem", "wait_nop", "branch"]
    rows = []
    rows.append(f"photo_kernel<{CS},{FS},true,{MODE}> -- instructions per phase (static = per 64-pixel wave slice inside the sub-tile loop)")
    rows.append(f"(VGPRs, VGPR spills, SGPR spills): marked build {vgprs(lines, CS, FS, MODE)}; unmarked build {vgprs(plain_lines, CS, FS, MODE)}")
    rows.append(f"{'phase':24s}" + "".join(f"{c:>10s}" for c in cols) + f"{'VALU all':>10s}")
    tot = Counter()
    loop = Counter()
    for ph, cnt in phases.items():
        va = cnt["valu"] + cnt["valu_pk"] + cnt["valu_dpp"] + cnt["valu_lane"]
        rows.append(f"{ph:24s}" + "".join(f"{cnt[c]:10d}" for c in cols) + f"{va:10d}")
        tot.update(cnt)
        if ph not in ("prologue", "loop_end", "B_texture_path"):
            loop.update(cnt)
    va = lambda c: c["valu"] + c["valu_pk"] + c["valu_dpp"] + c["valu_lane"]
    rows.append(f"{'sum (marked build)':24s}" + "".join(f"{tot[c]:10d}" for c in cols) + f"{va(tot):10d}")
    rows.append(f"{'staged slice (A..E)':24s}" + "".join(f"{loop[c]:10d}" for c in cols) + f"{va(loop):10d}")
    ptot = Counter()
    for cnt in plain.values():
        ptot.update(cnt)
    rows.append(f"{'sum (unmarked build)':24s}" + "".join(f"{ptot[c]:10d}" for c in cols) + f"{va(ptot):10d}")
    rows.append("")
    for ph in phases:
        if ph in ops:
            top = ", ".join(f"{o} {n}" for o, n in ops[ph].most_common(14))
            rows.append(f"{ph}: {top}")
    text = "\n".join(rows)
    print(text)
    if out_path:
        open(out_path, "w").write(text + "\n")


if __name__ == "__main__":
    main()
