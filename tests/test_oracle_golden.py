"""CPU: the oracle restatement (oracle/) against golden vectors produced by the reference itself."""
import json
import os
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden, golden_state, rel_inf, pkg

sys.path.insert(0, os.path.join(ROOT, "oracle"))
import denoiser_ref as R  # noqa: E402
import diffusion_ref as DR  # noqa: E402

FWD_CASES = ["fwd_tiny", "fwd_tiny_l2", "fwd_tiny_eqdim", "fwd_small_dims", "fwd_big_dims", "fwd_tools_shape"]
TOL = 2e-5  # fp32 vs fp32, different contraction order only


def test_layout_matches_reference_dump(golden_dir):
    lay = json.load(open(os.path.join(golden_dir, "state_dict_layout.json")))
    L = pkg("layout")
    for tag, rec in lay.items():
        ours = L.state_dict_layout(263, **rec["kwargs"])
        assert [[k, list(s)] for k, s in ours] == rec["keys"], tag
        n_params = sum(int(np.prod(s)) for k, s in ours if k.rsplit(".", 1)[-1] not in L.BUFFER_LEAVES)
        assert n_params == rec["n_params"], tag


@pytest.mark.parametrize("case", FWD_CASES)
def test_forward_matches_reference(case):
    g, meta = load_golden(case)
    sd, eph, proj, mcfg = golden_state(meta)
    trace = {}
    with torch.no_grad():
        y = R.denoiser_forward(sd, mcfg, g["x"], g["timesteps"], g["length"], g["xf_proj"], g["xf_out"], eph, proj,
                               trace=trace)
    assert y.shape == g["output"].shape
    for k, v in g.items():
        if not k.startswith("trace/"):
            continue
        name = k[len("trace/"):]
        if name.endswith(".moe.top2_idx"):
            ours = trace[name.replace(".moe.top2_idx", ".top2_idx")]
            assert torch.equal(ours, v), name
        elif name.endswith((".top2_val", ".gap23")):
            continue
        elif name in trace:
            assert rel_inf(trace[name], v) < TOL, name
    assert rel_inf(y, g["output"]) < TOL
    # MoE counters (switch_moe.py:71-92) start from zero in the fixture
    for k, v in g.items():
        if k.startswith("buf/") and k.endswith("expert_usage"):
            pre = k[len("buf/"):].replace(".moe.expert_usage", "")
            assert torch.equal(trace[pre + ".usage"], v)
        if k.startswith("buf/") and k.endswith("expert_importance"):
            pre = k[len("buf/"):].replace(".moe.expert_importance", "")
            assert torch.allclose(trace[pre + ".importance"], v, rtol=1e-4, atol=1e-4)


def test_projection_recipe_matches_reference():
    g, meta = load_golden("projection_qr")
    for dh in (16, 128):
        gen = torch.Generator().manual_seed(meta["seed"])
        P = R.create_projection(dh, 256, gen)
        assert P.shape == g[f"P{dh}"].shape == (dh, min(dh, 256))
        # QR sign/rounding is LAPACK-build dependent only at the 1e-6 level
        assert torch.allclose(P, g[f"P{dh}"], atol=1e-5)


def _loop_setup():
    g, meta = load_golden("loops_tiny")
    sd, eph, proj, mcfg = golden_state(meta)
    synth = pkg("synth")
    B, T, F_ = g["x_T"].shape

    def model(x, t, cond):
        xp, xo = (g["xf_proj"], g["xf_out"]) if cond else (g["xf_proj_uncond"], g["xf_out_uncond"])
        return R.denoiser_forward(sd, mcfg, x, t, g["length"], xp, xo, eph, proj)

    def noises(tag, n):
        return [synth.uniform_pm1((B, T, F_), f"noise.{tag}.{i}", meta["iseed"]) * (3.0 ** 0.5) for i in range(n)]

    return g, meta, model, noises


def test_tables_match_reference():
    g, meta = load_golden("loops_tiny")
    tb = DR.Tables(DR.linear_betas(meta["steps_cfg"]))
    for ours, name in [(tb.betas, "betas"), (tb.acp, "alphas_cumprod"), (tb.acp_prev, "alphas_cumprod_prev"),
                       (tb.sqrt_recip_acp, "sqrt_recip_alphas_cumprod"), (tb.sqrt_recipm1_acp, "sqrt_recipm1_alphas_cumprod"),
                       (tb.post_var, "posterior_variance"), (tb.post_logvar_clipped, "posterior_log_variance_clipped"),
                       (tb.coef1, "posterior_mean_coef1"), (tb.coef2, "posterior_mean_coef2")]:
        assert np.array_equal(ours, g["tables/" + name].numpy()), name
    tb = DR.Tables(DR.linear_betas(1000))
    for ours, name in [(tb.post_logvar_clipped, "posterior_log_variance_clipped"), (tb.coef1, "posterior_mean_coef1"),
                       (tb.coef2, "posterior_mean_coef2"), (tb.sqrt_recip_acp, "sqrt_recip_alphas_cumprod"),
                       (tb.sqrt_recipm1_acp, "sqrt_recipm1_alphas_cumprod")]:
        assert np.array_equal(ours, g["tables1000/" + name].numpy()), name


def test_cfg_loop_matches_reference():
    g, meta, model, noises = _loop_setup()
    tb = DR.Tables(DR.linear_betas(meta["steps_cfg"]))
    keep = []
    with torch.no_grad():
        y = DR.cfg_ddpm_loop(model, tb, g["x_T"], noises("cfg", meta["steps_cfg"]), cfg_scale=meta["cfg_scale"], keep=keep)
    for j, i in enumerate(g["cfg/traj_idx"].tolist()):
        assert rel_inf(keep[i], g["cfg/traj"][j]) < 2e-4, i
    assert rel_inf(y, g["cfg/final"]) < 2e-4


@pytest.mark.parametrize("eta", [0.0, 0.5])
def test_ddim_loop_matches_reference(eta):
    g, meta, model, noises = _loop_setup()
    tb = DR.Tables(DR.linear_betas(meta["steps_ddim"]))
    with torch.no_grad():
        y = DR.ddim_loop(model, tb, g["x_T"], noises(f"ddim.{eta}", meta["steps_ddim"]), eta=eta)
    assert rel_inf(y, g[f"ddim{eta}/final"]) < 2e-4


def test_text_head_oracle_matches_reference_golden():
    """oracle/text_head_ref.py against the reference's own EnhancedTextEncoder.forward output (DeBERTa fetch stubbed)."""
    import text_head_ref as TR
    g, meta = load_golden("text_head")
    sd = {k[3:]: v for k, v in g.items() if k.startswith("sd/")}
    pooled, projected = TR.text_head(g["hidden"], sd["prompt_tokens"], sd["proj.0.weight"], sd["proj.0.bias"],
                                     sd["proj.1.weight"], sd["proj.1.bias"])
    assert torch.equal(projected, g["projected"]) and torch.equal(pooled, g["pooled"])


def test_motion_post_oracle_matches_reference_golden():
    """oracle/motion_ref.py against the reference's recover_from_ric + motion_temporal_filter outputs."""
    import motion_ref as MR
    g, meta = load_golden("motion_post")
    for b, n in enumerate(g["length"].tolist()):
        j = MR.motion_to_joints(g["motion"][b, :n], g["mean"].numpy(), g["std"].numpy(), 22, 1.0)
        assert np.array_equal(j, g[f"joints/{b}"].numpy()), b
        r = MR.motion_to_joints(g["motion"][b, :n], g["mean"].numpy(), g["std"].numpy(), 22, 0.0)
        assert np.array_equal(r, g[f"joints_raw/{b}"].numpy()), b


def test_two_forward_cfg_with_different_token_counts_matches_reference_trainer():
    """The oracle's CFG loop with cond / uncond text of DIFFERENT token counts against the golden produced by the reference's
    own DDPMTrainer.generate (trainers/ddpm_trainer.py:145-199): the path a real tokenizer takes ("" -> fewer tokens)."""
    g, meta = load_golden("trainer_generate")
    sd, eph, proj, mcfg = golden_state(meta)
    synth = pkg("synth")
    Dt, Fe, steps, bs = meta["text_latent_dim"], meta["cfg"]["input_feats"], meta["steps"], meta["batch_size"]
    caps, lens = meta["captions"], g["m_lens"]
    tb = DR.Tables(DR.linear_betas(steps))

    def emb(c, N):
        return synth.uniform_pm1((N, Dt), "cap." + c, meta["iseed"]) * (3.0 ** 0.5)

    i_out = 0
    for k in range((len(caps) + bs - 1) // bs):
        lo, hi = k * bs, min((k + 1) * bs, len(caps))
        B = hi - lo
        T = min(int(lens[lo:hi].max()), meta["cfg"]["num_frames"])
        xo_c = torch.stack([emb(c, meta["N_cond"]) for c in caps[lo:hi]])
        xo_u = emb("", meta["N_uncond"])[None].expand(B, -1, -1).contiguous()

        def model(x, t, cond, xo_c=xo_c, xo_u=xo_u, ln=lens[lo:hi]):
            xo = xo_c if cond else xo_u
            return R.denoiser_forward(sd, mcfg, x, t, ln, xo.mean(1), xo, eph, proj)

        x_T = synth.uniform_pm1((B, T, Fe), f"gen.xT.{k}", meta["iseed"]) * (3.0 ** 0.5)
        nz = [synth.uniform_pm1((B, T, Fe), f"gen.noise.{k}.{i}", meta["iseed"]) * (3.0 ** 0.5) for i in range(steps)]
        keep = []
        with torch.no_grad():
            y = DR.cfg_ddpm_loop(model, tb, x_T, nz, cfg_scale=meta["cfg_scale"], keep=keep)
        if k == 0:
            for j, i in enumerate(g["traj0/idx"].tolist()):
                assert rel_inf(keep[i], g["traj0"][j]) < 2e-4, i
        for b in range(B):
            assert rel_inf(y[b], g[f"out/{i_out}"]) < 2e-4, i_out
            i_out += 1


def test_philox_known_answers():
    """oracle/philox_ref.py (the restatement of csrc/noise.hip) against the published Random123 known-answer vectors of
    philox4x32-10 (kat_vectors: zero, all-ones and the pi-digits counter/key)."""
    import philox_ref as P
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = P.philox4x32_10(np.array(ctr, dtype=np.uint32), key[0], key[1])
        assert tuple(int(v) for v in got) == want
    z = P.normal(4096, 8, 3, 99, 5)
    assert abs(float(z.mean())) < 0.02 and abs(float(z.std()) - 1) < 0.02
    assert np.array_equal(P.normal(4096, 2, 5, 99, 5), z[2:4])   # rows are a function of the GLOBAL sample index only
