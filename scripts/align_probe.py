"""dev tool (r06): the photometric / geometric linearize and the error pass on a window whose edges do NOT cut into a multiple of 8
runs (erode = 4 at 128x160: 17 168 samples = 68 sub-tiles = 9 runs of 8) -- with and without the XCD-aligned padding of the work list
(SAGE_XCD_ALIGN=0 / 1).  usage: python scripts/align_probe.py [erode]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sage_slam_amd import capi, synth
erode = int(sys.argv[1]) if len(sys.argv) > 1 else 4
w = synth.make_window(K=32, H=128, W=160, FS=16, CS=32, L=4, seed=0, erode=erode)
N = w.keyframes[0].homo.shape[0]
win = capi.Window(w)
win.linearize(); win.error(1); torch.cuda.synchronize()
win.set_profiling(True)
cfg = capi.lm_config_default(); cfg.max_inner_evals = 1; cfg.linearize_at_candidate = -1
for _ in range(6):
    win.reset(); win.lm_step(capi.SageLmState(), cfg)
mer = [win.kernel_time(i) for i in range(4)]
p = win.packed_host().astype("float64")
print(json.dumps({"N": int(N), "subtiles": (N + 255) // 256, "align": os.environ.get("SAGE_XCD_ALIGN", "1"),
                  "photo_merged_ms": round(mer[0][0] / max(1, mer[0][1]), 4), "geo_merged_ms": round(mer[1][0] / max(1, mer[1][1]), 4),
                  "error_pass_ms": round(mer[2][0] / max(1, mer[2][1]), 4), "packed_sum": float(p.sum())}))
