#!/usr/bin/env python3
"""Golden vectors of the reference's NearestPsd AS WRITTEN (core/mapping/mapping_utils.h:88-128: H = V^T diag(sigma) V
with Eigen's JacobiSVD matrixV(), then the LDLT / min-eigenvalue bump loop), produced by the reference's own source text
compiled against the vendored Eigen 3.3.9 (/root/reference/system/thirdparty/eigen).

The two function templates (IsPsd, NearestPsd) are cut out of the reference header at generation time into a scratch
file (the header itself cannot be included: it pulls gtsam / OpenCV / glog); the driver nearest_psd_golden_main.cpp
includes that scratch file.  Nothing of the reference is stored in this repository -- only the matrices it produced:
tests/golden/nearest_psd_eigen339.npz.  Build container only.

Usage:  python tests/golden/make_nearest_psd_golden.py
"""
import json
import os
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REF_HEADER = "/root/reference/system/sources/core/mapping/mapping_utils.h"
EIGEN = "/root/reference/system/thirdparty/eigen"


def main():
    lines = open(REF_HEADER).read().split("\n")
    start = next(i for i, l in enumerate(lines) if "inline bool IsPsd" in l) - 1          # its `template <typename T>` line
    end = next(i for i, l in enumerate(lines) if "return A3;" in l) + 1                   # closing brace of NearestPsd
    with tempfile.TemporaryDirectory() as tmp:
        open(os.path.join(tmp, "ref_nearest_psd.inc"), "w").write("\n".join(lines[start:end + 1]) + "\n")
        exe = os.path.join(tmp, "gen")
        subprocess.check_call(["g++", "-O2", "-std=c++14", "-I" + EIGEN, "-I" + tmp,
                               os.path.join(HERE, "nearest_psd_golden_main.cpp"), "-o", exe])
        out = subprocess.check_output([exe]).decode()
    import numpy as np
    cases = json.loads(out)
    arrs = {}
    for c in cases:
        n = c["n"]
        arrs[c["name"] + "_M"] = np.array(c["M"], np.float64).reshape(n, n)
        arrs[c["name"] + "_A"] = np.array(c["A"], np.float64).reshape(n, n)
        print(c["name"], n)
    np.savez_compressed(os.path.join(HERE, "nearest_psd_eigen339.npz"), **arrs)


if __name__ == "__main__":
    main()
