#!/usr/bin/env python3
"""Generate tests/golden/lm_trace_*.json: traces of the reference's tracker LM policy on BASELINE config 1
(2 keyframes, 64x80x16 feature maps, 32-dim code, N = 3072 samples + 160 matched keypoints).

The policy runs in oracle/track_lm.py -- an independent numpy restatement of camera_tracker.cpp:1156-1279 / :467-573 that
shares no code with the product's sage_track_lm -- over the C oracle's kernels (oracle/sage_oracle.c) composed as
CameraTracker::ComputeJacobianAndError / ComputeError do (tests/tracker_scene.py).  The cases cover: plain descent (6-dof
photometric + reprojection, 7-dof photometric + match geometry), a far start, an iteration whose candidates are all
REJECTED (damping climbs to max_damp, three inner evaluations), iterations that SKIP the Jacobian (update_jac == false,
provoked through jac_update_err_inc_threshold), accepted iterations on a
stale Jacobian and on a second inner evaluation (heavy damping), the max_num_iters exit and the no-overlap exit.

    python tests/golden/make_lm_trace_golden.py          (CPU only, ~1 min)
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as orc, track_lm as T      # noqa: E402
from tests.tracker_scene import HostScene              # noqa: E402

NEAR = ((0.004, -0.003, 0.002), (0.004, -0.003, 0.002))
FAR = ((0.02, -0.015, 0.01), (0.03, -0.02, 0.02))
# name, dof, use_photo, use_keypoints, (rot, trans) start offset, start scale factor, config overrides
CASES = [
    ("new_photo_reproj", 6, True, True, NEAR, 1.0, {}),
    ("frame_photo_matchgeom", 7, True, True, NEAR, 0.96, {}),
    ("new_photo_far", 6, True, False, FAR, 1.0, {}),
    ("frame_far_rejected", 7, True, True, FAR, 0.9, {}),
    ("frame_far_skipjac", 7, True, True, FAR, 0.9, {"jac_update_err_inc_threshold": 0.5}),
    ("new_far_skipjac", 6, True, True, FAR, 1.0, {"jac_update_err_inc_threshold": 0.9, "min_param_inc_thresh": 1e-4}),
    ("new_maxiter", 6, True, True, FAR, 1.0, {"max_num_iters": 2}),
    # heavy damping: short steps -> ACCEPTED iterations without a new Jacobian (prev_error then stays), an iteration that is
    # accepted on its second inner evaluation, and a final one that runs into max_damp
    ("new_damped_skipjac", 6, True, True, FAR, 1.0,
     {"init_damp": 2.0, "max_damp": 10.0, "jac_update_err_inc_threshold": 0.8, "damp_dec_factor": 3.0}),
    ("frame_damped_skipjac", 7, True, True, FAR, 0.9,
     {"init_damp": 2.0, "max_damp": 10.0, "jac_update_err_inc_threshold": 0.8, "damp_dec_factor": 3.0}),
    ("frame_no_overlap", 7, True, False, ((0.0, 0.0, 0.0), (50.0, 0.0, 0.0)), 1.0, {"no_overlap_error": "9.9*sum(photo_weights)"}),
]


def main():
    orc.build()
    sc = HostScene(orc)
    out_dir = os.path.dirname(os.path.abspath(__file__))
    for name, dof, up, uk, (rot, tr), s0f, over in CASES:
        cfg = T.default_config()
        over_num = dict(over)
        if over.get("no_overlap_error") == "9.9*sum(photo_weights)":
            over_num["no_overlap_error"] = float(np.float32(9.9) * np.float32(np.sum(sc.w.photo_weights)))
        cfg.update(over_num)
        lin, err = sc.oracle_callbacks(dof, up, uk)
        p0 = sc.start_pose(rot, tr)
        s0 = float(np.float32(float(sc.s_true) * s0f))
        pose, s, fe, it, trace, status = T.track_lm(cfg, dof, lin, err, p0, s0)
        rec = dict(name=name, dof=dof, use_photo=up, use_keypoints=uk, start_rot=list(rot), start_trans=list(tr),
                   start_pose=[float(v) for v in p0], start_scale=s0, config=over_num, status=status, iters=int(it),
                   final_error=float(fe), final_scale=float(s), final_pose=[float(v) for v in pose], trace=trace,
                   scene=dict(seed=31, NK=160))
        with open(os.path.join(out_dir, f"lm_trace_{name}.json"), "w") as f:
            json.dump(rec, f, indent=1)
        print(f"{name}: {status}, {it} iterations, {len(trace)} trace entries, "
              f"rejected {sum(1 for t in trace if not t['accepted'])}, skipped Jacobians {sum(1 for t in trace if not t['relinearized'])}")


if __name__ == "__main__":
    main()
