"""CPU: host-side logic of the drop-in layer (no kernels run): schedule tables, layouts, packing transforms,
module surface, loud failure without a GPU."""
import os
import types

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden, golden_state, pkg


def test_diffusion_tables_match_reference_golden():
    D = pkg("diffusion")
    g, meta = load_golden("loops_tiny")
    d = D.GaussianDiffusion(betas=D.get_named_beta_schedule("linear", meta["steps_cfg"]),
                            model_mean_type=D.ModelMeanType.EPSILON, model_var_type=D.ModelVarType.FIXED_SMALL,
                            loss_type=D.LossType.MSE)
    for name in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_recip_alphas_cumprod",
                 "sqrt_recipm1_alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped",
                 "posterior_mean_coef1", "posterior_mean_coef2"):
        assert np.array_equal(getattr(d, name), g["tables/" + name].numpy()), name
    tab = d.schedule_table()
    assert tab.shape == (7, meta["steps_cfg"]) and tab.dtype == np.float32
    assert np.array_equal(tab[2], g["tables/posterior_mean_coef1"].numpy().astype(np.float32))
    d1000 = D.GaussianDiffusion(betas=D.get_named_beta_schedule("linear", 1000), model_mean_type=D.ModelMeanType.EPSILON,
                                model_var_type=D.ModelVarType.FIXED_SMALL, loss_type=D.LossType.MSE)
    assert np.array_equal(d1000.posterior_mean_coef2, g["tables1000/posterior_mean_coef2"].numpy())


def test_unsupported_sampler_configs_raise():
    D = pkg("diffusion")
    d = D.GaussianDiffusion(betas=D.get_named_beta_schedule("linear", 50), model_mean_type=D.ModelMeanType.START_X,
                            model_var_type=D.ModelVarType.FIXED_SMALL, loss_type=D.LossType.MSE)
    with pytest.raises(NotImplementedError):
        d.p_sample_loop_with_cfg(None, (1, 2, 3))
    d = D.GaussianDiffusion(betas=D.get_named_beta_schedule("linear", 50), model_mean_type=D.ModelMeanType.EPSILON,
                            model_var_type=D.ModelVarType.LEARNED, loss_type=D.LossType.MSE)
    with pytest.raises(NotImplementedError):
        d.schedule_table()
    with pytest.raises(NotImplementedError):
        D.get_named_beta_schedule("nope", 10)


def test_module_state_dict_layout_and_cpu_refusal():
    """Same keys/shapes as the reference dump; forward on CPU tensors fails loudly (no fallback path)."""
    import json
    from conftest import GOLDEN
    T, L = pkg("transformer"), pkg("_lib")
    lay = json.load(open(os.path.join(GOLDEN, "state_dict_layout.json")))["tools_L2"]
    m = T.MotionTransformer(263, **lay["kwargs"])
    assert [[k, list(v.shape)] for k, v in m.state_dict().items()] == lay["keys"]
    assert sum(p.numel() for p in m.parameters()) == lay["n_params"]
    # reference init semantics: zero gates/out head, xavier'd Performer "zero" layers are NOT zero
    sd = m.state_dict()
    assert float(sd["out.weight"].abs().max()) == 0.0
    assert float(sd["decoder_blocks_low.0.module.ffn.branches.0.moe.gate.weight"].abs().max()) == 0.0
    assert float(sd["decoder_blocks_low.0.module.ffn.proj_out.out_layers.2.weight"].abs().max()) == 0.0
    assert float(sd["decoder_blocks_low.0.module.dual_self_attn.local_attn.style_block.out_layers.2.weight"].abs().max()) > 0
    x = torch.zeros(1, 4, 263)
    with pytest.raises(L.MdmError):
        m(x, torch.zeros(1, dtype=torch.long), torch.tensor([4]), xf_proj=torch.zeros(1, 128), xf_out=torch.zeros(1, 3, 128))
    with pytest.raises(L.MdmError):
        m.encode_text(["a"], "cpu")
    assert m.generate_src_mask(4, torch.tensor([2, 4])).tolist() == [[1, 1, 0, 0], [1, 1, 1, 1]]
    # load_state_dict(strict=False) tolerates the reference's extra text_encoder.* keys (ddpm_trainer.py:286-288)
    extra = dict(sd)
    extra["text_encoder.prompt_tokens"] = torch.zeros(1, 8, 1024)
    m.load_state_dict(extra, strict=False)


def test_kernel_layout_transforms_are_exact():
    """packing.kernel_layout: conv-as-linear weight reshapes reproduce F.conv1d / F.conv_transpose1d; stacking orders."""
    P, synth = pkg("packing"), pkg("synth")
    g, meta = load_golden("fwd_tiny")
    sd, eph, proj, mcfg = golden_state(meta)
    cfg = dict(mcfg, ff_size=128, text_latent_dim=32, input_feats=263, num_frames=16)
    lay = P.kernel_layout(sd, cfg, eph, proj)
    D = 64
    h = torch.randn(2, 8, D)
    ref = F.conv1d(h.permute(0, 2, 1), sd["downsample.weight"], sd["downsample.bias"], stride=2).permute(0, 2, 1)
    ours = h.reshape(2 * 4, 2 * D) @ lay["W:down"].T + lay["V:down_b"]
    assert torch.allclose(ours.reshape(2, 4, D), ref, atol=1e-5)
    hl = torch.randn(2, 4, D)
    ref = F.conv_transpose1d(hl.permute(0, 2, 1), sd["upsample.weight"], sd["upsample.bias"], stride=2).permute(0, 2, 1)
    ours = (hl.reshape(8, D) @ lay["W:up"].T + lay["V:up_b2"]).reshape(2, 8, D)
    assert torch.allclose(ours, ref, atol=1e-5)
    E, F_ = 4, 128
    assert lay["W:L0.w1"].shape == (2 * E * F_, D) and lay["W:L0.w2"].shape == (2 * E * D, F_)
    assert torch.equal(lay["W:L0.w1"][(1 * E + 2) * F_:(1 * E + 3) * F_],
                       sd["decoder_blocks_low.0.module.ffn.branches.1.moe.experts.2.0.weight"])
    assert torch.equal(lay["W:L1.local.qkv"][D:2 * D], sd["decoder_blocks_high.0.module.dual_self_attn.local_attn.key.weight"])
    Te = 4 * D
    assert lay["W:style_eph"].shape == (8 * Te, D) and lay["W:style_emb"].shape == (8 * 2 * D, Te)
    assert torch.equal(lay["W:style_eph"][2 * Te:3 * Te], eph["low.0.cross_style"][0])
    assert torch.equal(lay["W:style_emb"][7 * 2 * D:], sd["decoder_blocks_high.0.module.ffn.proj_out.emb_layers.1.weight"])
    assert torch.equal(lay["W:L0.global.feat"], proj["low.0.global"].t())


def test_synth_is_platform_exact_and_seeded():
    synth = pkg("synth")
    a = synth.uniform_pm1((4, 5), "x", 3)
    assert torch.equal(a, synth.uniform_pm1((4, 5), "x", 3)) and not torch.equal(a, synth.uniform_pm1((4, 5), "x", 4))
    assert float(a.abs().max()) < 1.0
    names = synth.ephemeral_names(2, True)
    assert names[0] == "text_proj" and len(names) == 1 + 2 * 2 * 4 and names[1] == "low.0.local_style"
    assert synth.synth_projection("p", 128, 0).shape == (128, 128)


def test_trainer_surface():
    Tr, T = pkg("trainer"), pkg("transformer")
    m = T.MotionTransformer(263, num_frames=8, latent_dim=64, ff_size=64, num_layers=1, num_heads=4, text_latent_dim=32,
                            moe_num_experts=2)
    args = types.SimpleNamespace(device=torch.device("cpu"), diffusion_steps=50, is_train=False)
    tr = Tr.DDPMTrainer(args, m)
    assert tr.cfg_scale == 7.5 and tr.diffusion.num_timesteps == 50
    assert tr._model() is m
    with pytest.raises(NotImplementedError):
        tr.train()


def test_trainer_checkpoint_roundtrip(tmp_path):
    """DDPMTrainer.save / load keep the reference's checkpoint dict (ddpm_trainer.py:260-289)."""
    Tr, T = pkg("trainer"), pkg("transformer")
    kw = dict(num_frames=8, latent_dim=64, ff_size=64, num_layers=1, num_heads=4, text_latent_dim=32, moe_num_experts=2)
    m1, m2 = T.MotionTransformer(263, **kw), T.MotionTransformer(263, **kw)
    args = types.SimpleNamespace(device=torch.device("cpu"), diffusion_steps=50, is_train=False)
    t1, t2 = Tr.DDPMTrainer(args, m1), Tr.DDPMTrainer(args, m2)
    f = str(tmp_path / "latest.tar")
    t1.save(f, ep=3, total_it=77)
    ck = torch.load(f)
    assert set(ck) == {"opt_encoder", "ep", "total_it", "encoder"}
    assert t2.load(f) == (3, 77)
    for (k1, v1), (k2, v2) in zip(m1.state_dict().items(), m2.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)


def test_elementwise_diffusion_helpers_match_oracle():
    """q_sample / posterior / eps<->x0 conversions (gaussian_diffusion.py:433-475,554-571) against the oracle's tables."""
    import importlib, os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    R = importlib.import_module("oracle.diffusion_ref")
    D = pkg("diffusion")
    steps = 50
    d = D.GaussianDiffusion(betas=D.get_named_beta_schedule("linear", steps), model_mean_type=D.ModelMeanType.EPSILON,
                            model_var_type=D.ModelVarType.FIXED_SMALL, loss_type=D.LossType.MSE)
    tb = R.Tables(R.linear_betas(steps))
    g = torch.Generator().manual_seed(3)
    x, eps, x0 = (torch.randn(3, 5, 7, generator=g) for _ in range(3))
    t = torch.tensor([0, 17, 49])
    f = lambda arr: torch.tensor([tb.f32(arr, int(i)) for i in t])[:, None, None]
    assert torch.equal(d._predict_xstart_from_eps(x, t, eps), f(tb.sqrt_recip_acp) * x - f(tb.sqrt_recipm1_acp) * eps)
    assert torch.equal(d._predict_eps_from_xstart(x, t, x0), (f(tb.sqrt_recip_acp) * x - x0) / f(tb.sqrt_recipm1_acp))
    mean, var, logvar = d.q_posterior_mean_variance(x0, x, t)
    assert torch.equal(mean, f(tb.coef1) * x0 + f(tb.coef2) * x)
    assert torch.equal(var, f(tb.post_var).expand_as(x)) and torch.equal(logvar, f(tb.post_logvar_clipped).expand_as(x))
    q = d.q_sample(x0, t, noise=eps)
    assert torch.equal(q, f(np.sqrt(tb.acp)) * x0 + f(np.sqrt(1.0 - tb.acp)) * eps)
    m, v, lv = d.q_mean_variance(x0, t)
    assert torch.equal(m, f(np.sqrt(tb.acp)) * x0) and torch.equal(v, f(1.0 - tb.acp).expand_as(x0))


def test_erf_rational_of_the_kernels_is_accurate():
    """The branch-free rational erf used by every GELU epilogue (csrc/mdm_common.h: erf_fast): its coefficients are read
    from the header and evaluated in fp32 exactly as the kernel does (Horner with fused multiply-adds emulated in
    fp64 -> fp32 rounding per step); max abs error against scipy's erf over [-6, 6] must stay below 5e-7."""
    import os, re
    from scipy.special import erf
    from conftest import ROOT
    src = open(os.path.join(ROOT, "motiondiffusion-moe_amd", "csrc", "mdm_common.h")).read()
    body = src[src.index("float erf_fast(float x)"):src.index("float gelu_erf(float x)")]
    nums = [float(v) for v in re.findall(r"(-?\d\.\d+e[-+]\d+)f", body)]
    assert len(nums) == 12, nums
    p_c, q_c = nums[:7], nums[7:]
    x = np.linspace(-6, 6, 200001).astype(np.float32)
    xc = np.clip(x, -4, 4)
    x2 = (xc * xc).astype(np.float32)

    def horner(cs):
        acc = np.full_like(x2, np.float32(cs[0]))
        for c in cs[1:]:
            acc = (acc.astype(np.float64) * x2.astype(np.float64) + np.float64(np.float32(c))).astype(np.float32)  # fmaf
        return acc

    approx = (xc * horner(p_c)).astype(np.float32) / horner(q_c)
    assert float(np.abs(approx.astype(np.float64) - erf(x.astype(np.float64))).max()) < 5e-7


def test_sigmoid_form_gelu_of_the_fused_mlp_is_accurate():
    """gelu_sig2 (csrc/mdm_common.h), the GELU of the streamed-weight expert MLP: x * sigmoid(q(x)) with the header's
    coefficients evaluated in fp32 as the kernel does; max abs error against the exact erf form < 5e-6 over [-12, 12] (dense),
    [-1e4, 1e4] and the decades up to +-1e12 (beyond the clamp at 6.5 the result is x * 8.7e-20 for negative x, an
    error linear in |x| that reaches the fit's own 3.5e-6 only at |x| ~ 4e13)."""
    import os, re
    from scipy.special import erf
    from conftest import ROOT
    src = open(os.path.join(ROOT, "motiondiffusion-moe_amd", "csrc", "mdm_common.h")).read()
    body = src[src.index("f32x2 gelu_sig2(f32x2 v)"):src.index("// exp on the hardware exp2 unit")]
    nums = [float(v) for v in re.findall(r"(-?\d\.\d+(?:e[-+]\d+)?)f", body)]
    clamp, cs = nums[:4], nums[4:14:2]
    assert clamp == [-6.5, 6.5, -6.5, 6.5] and len(cs) == 5, nums
    x = np.concatenate([np.linspace(-12, 12, 480001), np.linspace(-1e4, 1e4, 200001),
                        np.array([s * 10.0 ** k for k in range(5, 13) for s in (-1.0, 1.0)])]).astype(np.float32)
    xc = np.clip(x, -6.5, 6.5)
    x2 = (xc * xc).astype(np.float32)
    p = np.full_like(x2, np.float32(cs[0]))
    for c in cs[1:]:
        p = (p.astype(np.float64) * x2.astype(np.float64) + np.float64(np.float32(c))).astype(np.float32)
    e = np.exp2((p * xc).astype(np.float32)).astype(np.float32)
    got = (x / (np.float32(1) + e)).astype(np.float64)
    want = 0.5 * x.astype(np.float64) * (1 + erf(x.astype(np.float64) / np.sqrt(2.0)))
    assert float(np.abs(got - want).max()) < 5e-6


def test_load_balancing_loss_matches_reference_golden():
    """switch_moe.py:113-145 on given counters (tests/golden/moe_loss.npz, produced by the reference's own method)."""
    T = pkg("transformer")
    g, _ = load_golden("moe_loss")
    for E in (4, 8):
        got = T.MotionTransformer.load_balancing_loss(g[f"usage{E}"], g[f"importance{E}"])
        assert torch.allclose(got.reshape(1), g[f"loss{E}"], rtol=1e-6, atol=1e-7), (E, got, g[f"loss{E}"])
    z = torch.zeros(8)
    assert torch.allclose(T.MotionTransformer.load_balancing_loss(z, z).reshape(1), g["loss_zero_counters"])
