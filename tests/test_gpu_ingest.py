"""Caller side of the hot path (north_star: "the TorchScript feature/depth nets run once per keyframe via PyTorch-ROCm and
hand their outputs to the HIP path as plain device buffers"): scripted, randomly initialised stand-ins for
jit_feat_model.pt / jit_depth_model.pt (same output contracts: feature map [1,FS,H,W] in tanh range, depth bias [1,1,H,W] and
depth Jacobian w.r.t. the code [H*W,CS]) produce the keyframe inputs ON THE DEVICE; capi.keyframe_from_net_outputs builds the
keyframe bundle with the f1 producers (pyramid + gradients, valid pixels, seeded sampling) without a host round trip; a window
over those device tensors linearises bit-identically to a window fed the same arrays through host copies, and an edge matches
the CPU oracle.  (No checkpoints in this environment: the networks are random; this is the plumbing, not the accuracy.)"""
import types

import numpy as np
import pytest

from sage_slam_amd import synth
from tests.helpers import oracle_geo, oracle_photo, rel

pytestmark = pytest.mark.gpu


def _nets(FS, CS):
    import torch
    import torch.nn as nn

    class FeatNet(nn.Module):
        def __init__(self):
            super().__init__()
            self.c1 = nn.Conv2d(3, 24, 5, padding=2)
            self.c2 = nn.Conv2d(24, FS, 5, padding=2)

        def forward(self, x):
            return torch.tanh(self.c2(torch.relu(self.c1(x))))

    class DepthNet(nn.Module):
        def __init__(self):
            super().__init__()
            self.b = nn.Conv2d(3, 1, 7, padding=3)
            self.j = nn.Conv2d(3, CS, 7, padding=3)

        def forward(self, x):
            bias = 1.0 + 0.2 * torch.tanh(self.b(x))                                  # [1,1,H,W], positive
            jac = 0.05 * torch.tanh(self.j(x))                                        # [1,CS,H,W]
            return bias, jac.reshape(jac.shape[1], -1).transpose(0, 1).contiguous()   # [H*W,CS]

    torch.manual_seed(0)
    return torch.jit.script(FeatNet()).cuda().eval(), torch.jit.script(DepthNet()).cuda().eval()


def test_net_outputs_to_window_without_host_round_trip(orc):
    import torch
    assert torch.cuda.is_available()
    from sage_slam_amd import capi
    K, H, W, FS, CS, L, NS = 3, 64, 80, 16, 32, 4, 3072
    ref = synth.make_window(K=K, H=H, W=W, FS=FS, CS=CS, L=L, n_samples=NS, seed=3)   # cameras, mask, poses, weights
    feat_net, depth_net = _nets(FS, CS)
    ws = capi.Workspace()
    pyr = capi.make_pyramid(ref.cams[0], L)
    mask = torch.from_numpy(ref.mask).cuda()
    rng = np.random.default_rng(5)
    base = torch.from_numpy(rng.random((1, 3, H + 8, W + 8), dtype=np.float32)).cuda()
    kfs = []
    with torch.no_grad():
        for k in range(K):
            img = torch.nn.functional.avg_pool2d(base[:, :, k:k + H + 4, 2 * k:2 * k + W + 4], 5, stride=1, padding=0)
            fmap = feat_net(img)
            bias, jac = depth_net(img)
            assert fmap.shape == (1, FS, H, W) and bias.shape == (1, 1, H, W) and jac.shape == (H * W, CS)
            kfs.append(capi.keyframe_from_net_outputs(ws, fmap, bias, jac, mask, pyr, seed=1000 + k, num_samples=NS,
                                                      R=ref.keyframes[k].R, t=ref.keyframes[k].t))
    for kf in kfs:
        assert kf.feat_pyr.is_cuda and kf.basis.is_cuda and kf.loc1d.dtype == torch.int64 and kf.homo.shape == (NS, 3)
        assert kf.avg_squared_dpt_bias > 0
    dev = types.SimpleNamespace(**{f: getattr(ref, f) for f in ("H", "W", "L", "FS", "CS", "cams", "level_offsets", "P", "mask",
                                                                "links", "photo_weights", "geo_weight", "eps")})
    dev.geo_loss_param = 0.03 * float(np.mean([kf.avg_squared_dpt_bias for kf in kfs]))   # geo_loss_param_factor * avg bias^2
    dev.link_geo_loss = [0.03 * kfs[b].avg_squared_dpt_bias for a, b in ref.links]         # per link, from the newer keyframe
    dev.keyframes = kfs
    win = capi.Window(dev)
    assert all(dk.feat_pyr.data_ptr() == kf.feat_pyr.data_ptr() and dk.basis.data_ptr() == kf.basis.data_ptr()
               for dk, kf in zip(win.kfs, kfs)), "the window must use the producers' device buffers as they are"
    win.linearize()
    p_dev = win.packed_host().copy()
    # the same keyframes through host copies (the usual test path)
    host = types.SimpleNamespace(**vars(dev))
    host.keyframes = [synth.Keyframe(kf.feat_pyr.cpu().numpy(), kf.grad_pyr.cpu().numpy(), kf.bias.cpu().numpy(),
                                     kf.basis.cpu().numpy(), kf.code, kf.scale, kf.loc1d.cpu().numpy(),
                                     kf.homo.cpu().numpy(), kf.R, kf.t) for kf in kfs]
    win2 = capi.Window(host)
    win2.linearize()
    assert np.array_equal(win2.packed_host(), p_dev)
    # one link against the oracle (its own geo loss parameter)
    a, b = ref.links[0]
    for d, (k0, k1) in enumerate(((a, b), (b, a))):
        host.geo_loss_param = dev.link_geo_loss[0]
        for t, fn in ((0, oracle_photo), (1, oracle_geo)):
            o, h = fn(orc, host, k0, k1), win.get_edge(t, d)
            assert h["num_inliers"] == o["num_inliers"]
            assert rel(h["AtA"], o["AtA"]) < 2e-5 and rel(h["Atb"], o["Atb"]) < 2e-5, (t, d)
    cfg = capi.lm_config_default()
    st = capi.SageLmState()
    win.lm_step(st, cfg)
    assert np.isfinite(st.error) and np.isfinite(st.candidate_error)
    win.close(); win2.close(); ws.close()
