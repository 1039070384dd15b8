"""Build-time checks of the generated device code (CPU; needs hipcc, which cross-compiles without a GPU)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_hand_tracked_loads_are_waited_for_before_any_use():
    """The LDS-staged sampler of the photometric kernels issues loads in inline asm that the compiler's wait-count
    bookkeeping does not see (photo_kernels.hip: gload16 / vm_wait_keep).  The compiler may legally copy or spill such a
    register between the load and the hand-written wait -- reading it before the data has landed.  The checker walks the
    control-flow graph of the generated assembly from every such load to its covering wait."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "check_asm_loads.py")], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "0 problems" in r.stdout


def test_checker_flags_a_use_before_the_wait():
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import check_asm_loads as c
    bad = """_Zk:
	;;#ASMSTART
	s_nop 4
	global_load_dwordx4 v[4:7], v1, s[2:3]
	;;#ASMEND
	v_mov_b32_e32 v9, v5
	;;#ASMSTART
	s_waitcnt vmcnt(0)
	;;#ASMEND
	s_endpgm
"""
    n, problems = c.check(bad)
    assert n == 1 and len(problems) == 1 and "v_mov_b32" in problems[0][2]
    good = bad.replace("	v_mov_b32_e32 v9, v5\n", "	v_mov_b32_e32 v9, v10\n")
    n, problems = c.check(good)
    assert n == 1 and not problems
    # a path that branches around the wait is followed too
    branchy = """_Zk:
	;;#ASMSTART
	s_nop 4
	global_load_dwordx4 v[4:7], v1, s[2:3]
	;;#ASMEND
	s_cbranch_scc1 .LBB0_2
	;;#ASMSTART
	s_waitcnt vmcnt(0)
	;;#ASMEND
.LBB0_2:
	v_add_f32_e32 v0, v4, v4
	s_endpgm
"""
    n, problems = c.check(branchy)
    assert n == 1 and problems


def test_makefile_builds_the_same_translation_units_as_build_py():
    """The root Makefile (the no-Python build for a C++ maintainer) and sage_slam_amd/build.py must not drift apart: same
    sources, same headers in the dependency list, and `make -n` resolves every rule."""
    import re
    from sage_slam_amd import build as b
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "Makefile")).read().replace("\\\n", " ")
    hip = re.search(r"^HIP_SRC\s*:=\s*(.*)$", text, re.M).group(1).split()
    host = re.search(r"^HOST_SRC\s*:=\s*(.*)$", text, re.M).group(1).split()
    assert sorted(hip + host) == sorted(b.SOURCES)
    hdr = re.search(r"^HEADERS\s*:=\s*(.*)$", text, re.M).group(1)
    for h in b.HEADERS:
        assert os.path.basename(h) in hdr, h
    out = subprocess.run(["make", "-n", "-B", "OBJ=/tmp/_sage_mk_dry", "LIB=/tmp/_sage_mk_dry.so"], cwd=root,
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert out.stdout.count("--offload-arch=gfx950") == len(hip) + 1      # every .hip + the link line
    assert out.stdout.count("clang++") == len(host)


@pytest.mark.parametrize("cc,std,lang", [("gcc", "-std=c99", "c"), ("g++", "-std=c++11", "c++")])
def test_public_header_is_plain_c_abi(tmp_path, cc, std, lang):
    """include/sage_ba.h is the drop-in boundary: it must compile on its own as strict C99 (a cgo / ctypes / C caller) and as
    C++11 (the reference's translation units), warnings as errors, with nothing but the standard headers."""
    import shutil
    if not shutil.which(cc):
        pytest.skip(f"{cc} not installed")
    src = tmp_path / "hdr_only.c"
    src.write_text('#include "sage_ba.h"\nint main(void) { return (int)sizeof(SageCamera) == 0; }\n')
    r = subprocess.run([cc, std, "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + os.path.join(ROOT, "include"), "-x", lang,
                        "-c", str(src), "-o", str(tmp_path / "hdr_only.o")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
