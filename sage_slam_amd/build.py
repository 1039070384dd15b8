"""Build the HIP/C++ engine in-tree: sage_slam_amd/libsage_ba.so (gfx950 only).

hipcc cross-compiles without a GPU; the .so travels to the GPU box with the tree (it is git-ignored, not
gpurun-ignored).  Usage: ``python -m sage_slam_amd.build [--force]``.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "_obj")
LIB = os.path.join(HERE, "libsage_ba.so")
SOURCES = ["photo_kernels.hip", "geo_kernels.hip", "track_kernels.hip", "producers.hip", "keypoint_kernels.hip", "solve_kernels.hip", "operators.hip", "tracker.hip", "window.hip", "window_dist.hip", "window_factors.hip",
           "host_math.cpp", "shard_solve.cpp"]
HEADERS = ["sage_device.h", "sage_internal.h", "host_math.h", "runtime_internal.h", "finalize_bodies.h", os.path.join(ROOT, "include", "sage_ba.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
HOSTCXX = os.environ.get("HOSTCXX", "/opt/rocm/lib/llvm/bin/clang++")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]


def kernel_source_sha16() -> str:
    """Hash of the dense factor kernels' sources: recorded next to rocprofv3 counter summaries (scripts/pmc_summary.py) and
    compared by bench.py before it quotes committed per-launch counter constants -- instruction / byte counts belong to one
    build of these files."""
    import hashlib
    h = hashlib.sha256()
    for f in ("photo_kernels.hip", "geo_kernels.hip", "sage_device.h", "sage_internal.h"):
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def _mtime(p):
    return os.path.getmtime(p) if os.path.exists(p) else 0.0


def _compile(src):
    obj = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
    deps = [os.path.join(CSRC, src)] + [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]
    if _mtime(obj) >= max(_mtime(d) for d in deps):
        return obj, False
    if src.endswith(".hip"):
        cmd = [HIPCC] + FLAGS + ["-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj]
    else:   # pure host translation unit: ROCm's clang++ without offload (function multiversioning for AVX2/AVX-512)
        cmd = [HOSTCXX, "-O3", "-std=c++17", "-fPIC", "-Wall", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
               "-c", os.path.join(CSRC, src), "-o", obj]
    subprocess.check_call(cmd)
    return obj, True


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        res = list(ex.map(_compile, SOURCES))
    objs = [o for o, _ in res]
    if any(ch for _, ch in res) or not os.path.exists(LIB):
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-lpthread"])
        if verbose:
            print("built", LIB)
    return LIB


GLUE_SRC = os.path.join(ROOT, "integration", "compile_check", "prepass_run.cpp")
GLUE_BIN = os.path.join(ROOT, "integration", "compile_check", "_bin", "prepass_run")
REF_THIRDPARTY = "/root/reference/system/thirdparty"


def build_glue_check(verbose: bool = False):
    """Test binary that EXECUTES integration/sage_gtsam_prepass.h (f2 glue) against the engine library: real Eigen and
    Sophus (header-only, from the reference's thirdparty tree -> build container only), recording stand-ins for
    gtsam / boost.  The binary travels to the GPU box with the tree; tests/test_gpu_gtsam_glue.py runs it there.
    Returns the path, or None when the reference tree is absent."""
    if not os.path.isdir(REF_THIRDPARTY):
        return GLUE_BIN if os.path.exists(GLUE_BIN) else None
    deps = [GLUE_SRC, os.path.join(ROOT, "integration", "sage_gtsam_prepass.h"), os.path.join(ROOT, "include", "sage_ba.h"),
            os.path.join(ROOT, "integration", "compile_check", "gtsam", "linear", "HessianFactor.h"),
            os.path.join(ROOT, "integration", "compile_check", "gtsam", "nonlinear", "Values.h"), LIB]
    if _mtime(GLUE_BIN) >= max(_mtime(d) for d in deps):
        return GLUE_BIN
    os.makedirs(os.path.dirname(GLUE_BIN), exist_ok=True)
    subprocess.check_call([HOSTCXX, "-std=c++17", "-O1", "-w", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(ROOT, "integration", "compile_check"), "-I" + REF_THIRDPARTY + "/eigen",
                           "-I" + REF_THIRDPARTY + "/Sophus", "-I/opt/rocm/include", GLUE_SRC, "-o", GLUE_BIN,
                           "-L" + HERE, "-lsage_ba", "-L/opt/rocm/lib", "-lamdhip64",
                           "-Wl,-rpath,$ORIGIN/../../../sage_slam_amd", "-Wl,-rpath,/opt/rocm/lib"])
    if verbose:
        print("built", GLUE_BIN)
    return GLUE_BIN


ADAPTER_BIN = os.path.join(ROOT, "integration", "compile_check", "_bin", "adapter_run")
REF_SYSTEM = "/root/reference/system"


def build_adapter_run(verbose: bool = False):
    """Test binary that EXECUTES the two replacement translation units (integration/sage_adapter.cpp,
    sage_adapter_keypoints.cpp) behind the reference's OWN headers with PyTorch-ROCm's libtorch: the driver
    integration/compile_check/adapter_run.cpp builds at::Tensor arguments and calls all 7 + 11 `df::` entry points
    (VERDICT r5 item 3).  Needs the reference's headers -> build container only; the binary travels to the GPU box with the
    tree (same image: the libtorch it is linked against sits at the same path there), tests/test_gpu_adapter_run.py runs it.
    Returns the path, or None when the reference tree is absent and no binary was built earlier."""
    if not os.path.isdir(REF_SYSTEM):
        return ADAPTER_BIN if os.path.exists(ADAPTER_BIN) else None
    import torch
    T = os.path.dirname(torch.__file__)
    srcs = [os.path.join(ROOT, "integration", "sage_adapter.cpp"), os.path.join(ROOT, "integration", "sage_adapter_keypoints.cpp"),
            os.path.join(ROOT, "integration", "compile_check", "adapter_run.cpp")]
    deps = srcs + [os.path.join(ROOT, "include", "sage_ba.h"), LIB]
    if _mtime(ADAPTER_BIN) >= max(_mtime(d) for d in deps):
        return ADAPTER_BIN
    os.makedirs(os.path.dirname(ADAPTER_BIN), exist_ok=True)
    objdir = os.path.join(os.path.dirname(ADAPTER_BIN), "_adapter_obj")
    os.makedirs(objdir, exist_ok=True)
    inc = ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "integration", "compile_check"),
           "-I" + REF_SYSTEM + "/sources/cuda", "-I" + REF_SYSTEM + "/sources/common", "-I" + REF_SYSTEM + "/thirdparty/eigen",
           "-I" + T + "/include", "-I" + T + "/include/torch/csrc/api/include", "-I/opt/rocm/include"]
    flags = ["-x", "c++", "-std=c++17", "-O1", "-fPIC", "-w", "-D__HIP_PLATFORM_AMD__", "-DUSE_ROCM", "-DDF_CODE_SIZE=32",
             "-DDF_FEAT_SIZE=16", "-D_GLIBCXX_USE_CXX11_ABI=" + str(int(torch._C._GLIBCXX_USE_CXX11_ABI))]

    def cc(src):
        obj = os.path.join(objdir, os.path.splitext(os.path.basename(src))[0] + ".o")
        subprocess.check_call([HIPCC] + flags + inc + ["-c", src, "-o", obj])
        return obj

    with ThreadPoolExecutor(max_workers=3) as ex:
        objs = list(ex.map(cc, srcs))
    # torch/lib first: it bundles its own libamdhip64 (same SONAME as /opt/rocm's) -- one HIP runtime for libtorch and the
    # engine library, like capi.lib() arranges it for the Python harness
    # ONE HIP runtime in the process: torch bundles its own (file libamdhip64.so, SONAME libamdhip64.so.7), the engine library
    # asks for libamdhip64.so.7.  The loader walks the dependencies breadth-first in link order: libtorch_hip (and with it
    # torch's runtime) is listed BEFORE libsage_ba, whose request is then answered by the already loaded SONAME -- what
    # capi.lib() arranges for the Python harness by importing torch first.  (No direct -lamdhip64 here: it would bind
    # /opt/rocm's copy next to torch's.)
    subprocess.check_call([HOSTCXX] + objs + ["-o", ADAPTER_BIN, "-Wl,--no-as-needed", "-L" + T + "/lib", "-ltorch", "-ltorch_cpu",
                                              "-ltorch_hip", "-lc10", "-lc10_hip", "-L" + HERE, "-lsage_ba", "-lpthread",
                                              "-Wl,-rpath," + T + "/lib", "-Wl,-rpath,$ORIGIN/../../../sage_slam_amd",
                                              "-Wl,-rpath,/opt/rocm/lib"])
    if verbose:
        print("built", ADAPTER_BIN)
    return ADAPTER_BIN


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
