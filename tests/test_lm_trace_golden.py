"""a8: the product's tracker LM policy (sage_track_lm, csrc/host_math.cpp) against traces of an INDEPENDENT restatement of
the reference's loop (oracle/track_lm.py: camera_tracker.cpp:1156-1279, :467-512, :527-573; no shared code), committed as
tests/golden/lm_trace_*.json by tests/golden/make_lm_trace_golden.py.  Both sides evaluate through the C oracle's kernels on
BASELINE config 1, so what is compared is the policy: damping sequence, accept / reject, update_jac, iteration counts
EXACTLY; errors to 2e-5 -- the two sides solve the damped 6x6 / 7x7 systems with different column-pivoted QR codes (Eigen's
procedure restated vs LAPACK), which moves the iterates by fp32 rounding times the system's condition (measured <= 1e-5).
The `-m gpu` half (tests/test_gpu_tracker.py::test_track_frame_matches_golden_lm_trace) compares sage_track_frame -- the
same policy over the HIP kernels -- with the same files."""
import glob
import json
import os

import numpy as np
import pytest

from sage_slam_amd import capi
from tests.tracker_scene import HostScene

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "lm_trace_*.json")))


@pytest.fixture(scope="module")
def scene(orc):
    return HostScene(orc)


def product_config(rec):
    cfg = capi.lm_config_default()
    for k, v in rec["config"].items():
        setattr(cfg, k, v)
    return cfg


def check_trace(got, want, rtol):
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert np.float32(g["damp"]) == np.float32(w["damp"])
        assert g["accepted"] == w["accepted"] and g["relinearized"] == w["relinearized"]
        assert g["error"] == pytest.approx(w["error"], rel=rtol)
        assert g["candidate_error"] == pytest.approx(w["candidate_error"], rel=rtol)


def test_golden_set_covers_the_policy_branches():
    recs = [json.load(open(f)) for f in GOLDEN]
    assert len(recs) >= 10
    tr = [t for r in recs for t in r["trace"]]
    assert any(not t["accepted"] and t["inner_evals"] >= 3 for t in tr)               # damping climbed to max_damp
    assert any(t["accepted"] and not t["relinearized"] for t in tr)                   # accepted on a stale Jacobian
    assert any(t["accepted"] and t["inner_evals"] == 2 for t in tr)                   # accepted after one rejection
    assert any(r["status"] == "no_overlap" for r in recs)
    assert any(r["iters"] == r["config"].get("max_num_iters") for r in recs)
    assert {r["dof"] for r in recs} == {6, 7}


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[9:-5] for p in GOLDEN])
def test_sage_track_lm_walks_the_golden_trace(scene, path):
    rec = json.load(open(path))
    lin, err = scene.oracle_callbacks(rec["dof"], rec["use_photo"], rec["use_keypoints"])
    p0 = np.array(rec["start_pose"], np.float32)
    assert np.array_equal(p0, scene.start_pose(rec["start_rot"], rec["start_trans"]))   # the scene is the generator's
    cfg = product_config(rec)
    if rec["status"] == "no_overlap":
        with pytest.raises(capi.SageError) as ei:
            capi.track_lm(cfg, rec["dof"], lin, err, p0, rec["start_scale"])
        assert ei.value.code == -5                                                       # SAGE_E_NO_OVERLAP
        return
    pose, s, fe, it, tr = capi.track_lm(cfg, rec["dof"], lin, err, p0, rec["start_scale"])
    assert it == rec["iters"]
    check_trace(tr, rec["trace"], 2e-5)
    assert fe == pytest.approx(rec["final_error"], rel=2e-5)
    assert s == pytest.approx(rec["final_scale"], rel=1e-5)
    assert np.abs(pose - np.array(rec["final_pose"], np.float32)).max() < 2e-6
