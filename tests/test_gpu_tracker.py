"""a8: the tracker LM loops with the reference's term composition, HIP-wired (sage_track_frame) against the SAME policy
(sage_track_lm) driven by the CPU oracle's kernels composed in Python exactly as CameraTracker::ComputeJacobianAndError /
ComputeError do (camera_tracker.cpp:220-374):

  TrackNewFrame (dof 6):  photometric + reprojection
  TrackFrame    (dof 7):  photometric with scale + match geometry with scale; depths are handed over UNSCALED and every
                          evaluation multiplies them by the scale being evaluated (:264, :273, :431, :453)

BASELINE config 1 shape: 2 keyframes, 64x80x16 feature maps, 32-dim code, N = 3072 seeded samples.
"""
import numpy as np
import pytest

from sage_slam_amd import synth
from tests.helpers import rel
from tests.tracker_scene import HostScene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from sage_slam_amd import capi as c
    c.lib()
    return c


class Scene(HostScene):
    """the host scene + its device buffers and the product-side problem description"""

    def __init__(self, capi, orc, **kw):
        import torch
        super().__init__(orc, **kw)
        self.capi = capi
        w, a, b = self.w, self.a, self.b
        # device side
        self.ws = capi.Workspace()
        self.pyr = capi.make_pyramid(w.cams[0], w.L)
        dv = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
        self.dev = dict(mask=dv(w.mask), homo=dv(a.homo), f0s=dv(self.feat0s), feat1=dv(b.feat_pyr), grad1=dv(b.grad_pyr),
                        wts=dv(w.photo_weights), unscaled=dv(self.unscaled), metric=dv(self.s_true * self.unscaled),
                        kp_homo0=dv(self.kp_homo0), kp_unscaled=dv(self.kp_unscaled),
                        kp_metric=dv(self.s_true * self.kp_unscaled), kp_2d=dv(self.kp_matched_2d),
                        kp_dpts1=dv(self.kp_dpts1), kp_homo1=dv(self.kp_homo1))

    def problem(self, dof, use_photo, use_kp):
        capi, d, w = self.capi, self.dev, self.w
        p = capi.SageTrackProblem()
        p.ws = self.ws.h
        p.use_photo = int(use_photo)
        p.mask1_dev = d["mask"].data_ptr(); p.homo_dev = d["homo"].data_ptr(); p.feat0s_dev = d["f0s"].data_ptr()
        p.feat1_dev = d["feat1"].data_ptr(); p.grad1_dev = d["grad1"].data_ptr(); p.weights_dev = d["wts"].data_ptr()
        p.dpts0_dev = (d["unscaled"] if dof == 7 else d["metric"]).data_ptr()
        p.pyr = self.pyr; p.eps = w.eps; p.N = self.a.homo.shape[0]; p.FS = w.FS
        p.use_keypoints = int(use_kp); p.NK = self.NK
        p.kp_dpts0_dev = (d["kp_unscaled"] if dof == 7 else d["kp_metric"]).data_ptr()
        p.kp_homo0_dev = d["kp_homo0"].data_ptr(); p.kp_matched_2d_dev = d["kp_2d"].data_ptr()
        p.kp_matched_dpts1_dev = d["kp_dpts1"].data_ptr(); p.kp_matched_homo1_dev = d["kp_homo1"].data_ptr()
        p.kp_loss_param = self.reproj_loss_param if dof == 6 else self.mg_loss_param
        p.kp_weight = self.reproj_weight if dof == 6 else self.mg_weight
        return p

    def close(self):
        self.ws.close()


@pytest.fixture(scope="module")
def scene(capi, orc):
    s = Scene(capi, orc)
    yield s
    s.close()


@pytest.mark.parametrize("dof,use_photo,use_kp", [(6, True, False), (6, True, True), (6, False, True),
                                                  (7, True, False), (7, True, True), (7, False, True)],
                         ids=["new_photo", "new_photo+reproj", "new_reproj", "frame_photo", "frame_photo+matchgeom",
                              "frame_matchgeom"])
def test_track_frame_matches_oracle_wired_lm(capi, scene, dof, use_photo, use_kp):
    cfg = capi.lm_config_default()
    pose0 = scene.start_pose()
    s0 = float(scene.s_true) * (0.96 if dof == 7 else 1.0)   # low: LMConvergence uses the SIGNED max increment
    lin, err = scene.oracle_callbacks(dof, use_photo, use_kp)
    po, so, eo, ito, tro = capi.track_lm(cfg, dof, lin, err, pose0, s0)
    rc, ph, sh, eh, ith, trh = capi.track_frame(cfg, dof, scene.problem(dof, use_photo, use_kp), pose0, s0)
    assert rc == 0
    print(f"dof {dof} photo {use_photo} kp {use_kp}: iters {ith}/{ito}  error {tro[0]['error']:.5f} -> {eh:.5f}/{eo:.5f}  "
          f"scale {s0:.5f} -> {sh:.5f}/{so:.5f} (true {float(scene.s_true):.5f})  pose rel {rel(ph, po):.1e}")
    assert ith == ito and len(trh) == len(tro)
    assert [t["accepted"] for t in trh] == [t["accepted"] for t in tro]
    assert [t["relinearized"] for t in trh] == [t["relinearized"] for t in tro]
    np.testing.assert_allclose([t["error"] for t in trh], [t["error"] for t in tro], rtol=2e-4)
    np.testing.assert_allclose([t["candidate_error"] for t in trh], [t["candidate_error"] for t in tro], rtol=2e-4)
    assert eh == pytest.approx(eo, rel=1e-3)
    assert rel(ph, po) < 1e-4
    if dof == 7:
        # photometric term alone: the scale trades against the translation exactly, the damped 7x7 system is numerically
        # singular along that gauge and Eigen's colPivHouseholderQr().solve() KEEPS the tiny pivot (nonzeroPivots(), pinned
        # by tests/golden/colpiv_qr_eigen339.json; the rank()-style cut of round 2 dropped it) -- the step along the gauge
        # is then fp32 noise amplified by 1 / pivot in the reference as well; with a keypoint term the scale is observable
        assert sh == pytest.approx(so, rel=1e-4 if use_kp else 2e-3)
        # the scale is a live variable of the cost (it used to drift without moving the depths): started 4 %
        # low, the LM pulls it back towards the true scale
        # (only the match-geometry term sees the scale; the photometric term trades s against t exactly)
        if use_kp:
            assert abs(sh / float(scene.s_true) - 1.0) < 0.5 * 0.04
    assert eo < 0.7 * tro[0]["error"]                       # the LM actually descended
    if use_photo and (dof == 6 or use_kp):                  # (dof 7 photo-only may trade t against s; the keypoint-only
                                                            #  runs start inside their own noise: 0.4 px matches)
        assert np.linalg.norm(ph[9:] - scene.t10) < np.linalg.norm(pose0[9:] - scene.t10)


def test_track_frame_scale_changes_the_cost(capi, scene):
    """dof 7: two candidate scales give different errors through the product path (the error pass rescales the depths),
    equal to the oracle at the same scaled depths."""
    import ctypes as C
    cfg = capi.lm_config_default(); cfg.max_num_iters = 1
    prob = scene.problem(7, True, True)
    errs = []
    for s in (float(scene.s_true), 1.05 * float(scene.s_true)):
        rc, _, _, _, _, tr = capi.track_frame(cfg, 7, prob, scene.start_pose(), s)
        assert rc == 0 and len(tr) >= 1
        lin, _ = scene.oracle_callbacks(7, True, True)
        assert tr[0]["error"] == pytest.approx(lin(scene.start_pose(), s)[2], rel=2e-5)
        errs.append(tr[0]["error"])
    assert abs(errs[1] - errs[0]) > 1e-3 * errs[0]


def test_track_frame_no_overlap_exit(capi, scene):
    """TrackFrame without the match-geometry term: error >= 9.9 * sum(photo weights) ends the tracking with a failure
    (camera_tracker.cpp:1515-1519); with the keypoint term enabled the reference keeps going."""
    cfg = capi.lm_config_default()
    cfg.no_overlap_error = 9.9 * float(np.sum(scene.w.photo_weights))
    far = scene.start_pose().copy(); far[9:] += np.array([50.0, 0, 0], np.float32)     # warps everything out of the image
    rc, p, s, e, it, tr = capi.track_frame(cfg, 7, scene.problem(7, True, False), far, float(scene.s_true))
    assert rc == -5 and it == 0 and np.array_equal(p, far)                              # SAGE_E_NO_OVERLAP, pose untouched
    assert e == pytest.approx(10.0 * float(np.sum(scene.w.photo_weights)))             # the zero-inlier fallback value
    cfg.no_overlap_error = 0.0                                                          # use_match_geom: check is off
    rc, p, s, e, it, tr = capi.track_frame(cfg, 7, scene.problem(7, True, True), far, float(scene.s_true))
    assert rc == 0


def test_track_frame_rejects_bad_problems(capi, scene):
    cfg = capi.lm_config_default()
    rc = capi.track_frame(cfg, 7, scene.problem(7, False, False), scene.start_pose(), 1.0)[0]
    assert rc == -1                                                                     # "at least one factor should be enabled"
    rc = capi.track_frame(cfg, 5, scene.problem(6, True, False), scene.start_pose(), 1.0)[0]
    assert rc == -1


# ---------------------------------------------------------------------------------------------------------------------
# a8, independent oracle: sage_track_frame -- the product's policy over the HIP kernels -- against
# the committed traces of oracle/track_lm.py (a restatement of camera_tracker.cpp's loop that shares no code with
# sage_track_lm) over the C oracle's kernels: tests/golden/lm_trace_*.json, tests/golden/make_lm_trace_golden.py
# ---------------------------------------------------------------------------------------------------------------------
import glob
import json
import os

_GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "lm_trace_*.json")))


@pytest.mark.parametrize("path", _GOLDEN, ids=[os.path.basename(p)[9:-5] for p in _GOLDEN])
def test_track_frame_matches_golden_lm_trace(capi, scene, path):
    rec = json.load(open(path))
    cfg = capi.lm_config_default()
    for k, v in rec["config"].items():
        setattr(cfg, k, v)
    p0 = np.array(rec["start_pose"], np.float32)
    rc, ph, sh, eh, ith, trh = capi.track_frame(cfg, rec["dof"], scene.problem(rec["dof"], rec["use_photo"], rec["use_keypoints"]),
                                                p0, rec["start_scale"])
    if rec["status"] == "no_overlap":
        assert rc == -5 and ith == 0 and np.array_equal(ph, p0)
        return
    assert rc == 0 and ith == rec["iters"] and len(trh) == len(rec["trace"])
    for g, w in zip(trh, rec["trace"]):
        assert np.float32(g["damp"]) == np.float32(w["damp"])                        # the damping sequence, exactly
        assert g["accepted"] == w["accepted"] and g["relinearized"] == w["relinearized"]
        assert g["error"] == pytest.approx(w["error"], rel=2e-4)                     # HIP kernels vs the fp32 oracle's
        assert g["candidate_error"] == pytest.approx(w["candidate_error"], rel=2e-4)
    assert eh == pytest.approx(rec["final_error"], rel=1e-3)
    assert rel(ph, np.array(rec["final_pose"], np.float32)) < 1e-4
    if rec["dof"] == 7:
        assert sh == pytest.approx(rec["final_scale"], rel=1e-4)
