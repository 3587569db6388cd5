"""GPU: mdm_motion_postprocess against the outputs of the reference's recover_from_ric / motion_temporal_filter
(tests/golden/motion_post.npz, oracle/make_golden.py::case_motion_post) and the oracle restatement at full size."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden, pkg, rel_inf

pytestmark = pytest.mark.gpu
TOL = 2e-6  # fp32; only device sinf/cosf vs the host libm differ (everything else keeps the reference's rounding)


def test_joints_match_reference_golden():
    P = pkg("postprocess")
    g, meta = load_golden("motion_post")
    out = P.motion_to_joints(g["motion"].cuda(), g["mean"].numpy(), g["std"].numpy(), g["length"].cuda(), 22, sigma=1.0).cpu()
    raw = P.motion_to_joints(g["motion"].cuda(), g["mean"].numpy(), g["std"].numpy(), g["length"].cuda(), 22, sigma=0.0).cpu()
    for b, n in enumerate(g["length"].tolist()):
        assert rel_inf(raw[b, :n], g[f"joints_raw/{b}"]) < TOL, b
        assert rel_inf(out[b, :n], g[f"joints/{b}"]) < TOL, b
        assert torch.all(out[b, n:] == 0)


def test_full_size_against_oracle_and_taps():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import motion_ref as MR
    from scipy.ndimage import gaussian_filter1d
    P = pkg("postprocess")
    # the taps are scipy's kernel
    imp = np.zeros(21)
    imp[10] = 1.0
    for sigma in (1.0, 2.5):
        w = P.gaussian_taps(sigma)
        k = gaussian_filter1d(imp, sigma, mode="constant")
        r = len(w) - 1
        assert np.allclose(k[10:10 + r + 1], w, rtol=0, atol=1e-15)
    g = torch.Generator().manual_seed(5)
    B, T = 4, 196
    motion = torch.randn(B, T, 263, generator=g)
    mean = (torch.randn(263, generator=g) * 0.1).numpy()
    std = (0.5 + torch.rand(263, generator=g)).numpy()
    lengths = torch.tensor([196, 120, 41, 1])
    out = P.motion_to_joints(motion.cuda(), mean, std, lengths.cuda(), 22, sigma=1.0).cpu()
    for b, n in enumerate(lengths.tolist()):
        ref = MR.motion_to_joints(motion[b, :n], mean, std, 22, 1.0)
        assert rel_inf(out[b, :n], torch.from_numpy(ref)) < 5e-6, b
    # recover_from_ric with the reference's name on de-normalised data, no lengths, no filter
    data = motion[:2] * torch.from_numpy(std) + torch.from_numpy(mean)
    j = P.recover_from_ric(data.cuda(), 22).cpu()
    assert rel_inf(j, MR.recover_from_ric(data, 22)) < 5e-6


def test_bad_arguments_are_refused():
    L, P = pkg("_lib"), pkg("postprocess")
    with pytest.raises(L.MdmError):
        P.motion_to_joints(torch.zeros(1, 4, 263), np.zeros(263), np.ones(263))  # CPU tensor
    with pytest.raises(ValueError):
        P.motion_to_joints(torch.zeros(1, 4, 263, device="cuda"), np.zeros(10), np.ones(10))
    with pytest.raises(L.MdmError):
        P.motion_to_joints(torch.zeros(1, 4, 20, device="cuda"), np.zeros(20), np.ones(20))  # too few features for 22 joints
