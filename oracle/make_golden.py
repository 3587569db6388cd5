#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own modules (this container only).

    python oracle/make_golden.py            # needs /root/reference; never runs on the GPU box

Recipe: import /root/reference/text2motion/models with the DeBERTa text encoder
stubbed (its weights are a network fetch, unavailable offline), load synthetic weights
(motiondiffusion-moe_amd/synth.py -- regenerated from the seed in tests, so only inputs/outputs are stored),
inject the reference's per-call randomness explicitly:
  * the per-forward random nn.Linear layers (stylization.py:22-24, transformer.py:313-315) by patching
    nn.Linear.reset_parameters to copy from an ordered queue,
  * the lazily drawn Performer matrices (fast_attention.py:33-36) by setting the attribute,
  * randn_like in the sampling loops by patching torch.randn_like with a queue,
and record inputs, outputs and per-submodule activations.  Reference source is only imported, never copied.
"""
from __future__ import annotations

import contextlib
import importlib
import io
import json
import os
import sys

import numpy as np

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/text2motion"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

synth = importlib.import_module("motiondiffusion-moe_amd.synth")

with contextlib.redirect_stdout(io.StringIO()):
    import models.transformer as RT  # noqa: E402
    import models.gaussian_diffusion as RG  # noqa: E402
    import models.switch_moe as RS  # noqa: E402


class _StubTextEncoder(nn.Module):
    """Stands in for EnhancedTextEncoder (text_encoder.py:6-43): returns a fixed embedding per call."""

    table = {}

    def __init__(self, output_dim, dropout=0.1):
        super().__init__()

    def forward(self, text, device):
        key = "uncond" if all(t == "" for t in text) else "cond"
        xp, xo = _StubTextEncoder.table[key]
        return xp.clone(), xo.clone()


RT.EnhancedTextEncoder = _StubTextEncoder
OUT = os.path.join(ROOT, "tests", "golden")


class _EphemeralQueue:
    """Feeds nn.Linear() constructions inside forward from an ordered list."""

    def __init__(self, items):
        self.items = items
        self.i = 0
        self.orig = nn.Linear.reset_parameters

    def __enter__(self):
        q = self

        def reset(lin):
            n, w, b = q.items[q.i % len(q.items)]
            assert tuple(lin.weight.shape) == tuple(w.shape), (n, lin.weight.shape, w.shape)
            with torch.no_grad():
                lin.weight.copy_(w)
                lin.bias.copy_(b)
            q.i += 1

        nn.Linear.reset_parameters = reset
        return self

    def __exit__(self, *a):
        nn.Linear.reset_parameters = self.orig


def build_reference(cfg, wseed):
    with contextlib.redirect_stdout(io.StringIO()):
        m = RT.MotionTransformer(
            cfg["input_feats"], num_frames=cfg["num_frames"], latent_dim=cfg["latent_dim_arg"],
            ff_size=cfg["ff_size_arg"], num_layers=cfg["num_layers"], num_heads=cfg["num_heads"],
            text_latent_dim=cfg["text_latent_dim_arg"], moe_num_experts=cfg["moe_num_experts"],
            model_size=cfg["model_size"]).eval()
    sd = m.state_dict()
    new = synth.synth_state_dict([(k, tuple(v.shape)) for k, v in sd.items()], wseed)
    m.load_state_dict(new, strict=True)
    D = m.latent_dim
    dh = D // cfg["num_heads"]
    projs = synth.synth_projections(dh, cfg["num_layers"], wseed)
    fas = [mod for mod in m.modules() if type(mod).__name__ == "FastAttention"]
    assert len(fas) == len(projs)
    # module order == low.0.local, low.0.global, ..., high.*  (transformer.py:228-254 appends low,high per i)
    order = {}
    for name, mod in m.named_modules():
        if type(mod).__name__ == "FastAttention":
            parts = name.split(".")
            scale = "low" if "decoder_blocks_low" in name else "high"
            idx = parts[1]
            which = "local" if "local_attn" in name else "global"
            order[f"{scale}.{idx}.{which}"] = mod
    for n, p in projs:
        order[n].projection_matrix = p.clone()
    return m, D


def case_forward(name, cfg, B, T, N, wseed, iseed, trace_level="all"):
    m, D = build_reference(cfg, wseed)
    Dt = m.gated_fusion.proj_text.in_features if False else None
    Dt = cfg["text_latent_dim_arg"] * (2 if cfg["model_size"] == "big" else 1)
    eph = synth.synth_ephemerals(D, Dt, cfg["num_layers"], wseed)
    x, t, length, xf_proj, xf_out = synth.synth_inputs(B, T, cfg["input_feats"], N, Dt, iseed,
                                                       num_steps=1000, min_len=4)
    if cfg.get("odd_length"):
        length[-1] = max(1, T - 3)
    trace = {}
    hooks = []

    def mk(nm):
        def h(mod, inp, out):
            trace[nm] = out.detach().clone()
        return h

    def mk_moe(nm):
        def h(mod, inp, out):
            xf = inp[0].reshape(-1, inp[0].shape[-1])
            probs = torch.softmax(mod.gate(xf), dim=1)
            v, i = torch.topk(probs, k=2, dim=1)
            trace[nm + ".top2_idx"] = i.clone()
            trace[nm + ".top2_val"] = v.clone()
            srt = torch.sort(probs, dim=1, descending=True).values
            trace[nm + ".gap23"] = (srt[:, 1] - srt[:, 2]).clone()
        return h

    for nm, mod in m.named_modules():
        leaf = nm.split(".")[-1]
        if leaf in ("dual_self_attn", "cross_attn", "ffn", "sd_cross_attn", "local_attn", "global_attn"):
            hooks.append(mod.register_forward_hook(mk(nm)))
        if isinstance(mod, RS.SwitchMoELayer):
            hooks.append(mod.register_forward_hook(mk_moe(nm)))
    hooks.append(m.gated_fusion.register_forward_hook(mk("fused_emb")))
    hooks.append(m.downsample.register_forward_hook(lambda mod, i, o: trace.__setitem__("h_low", o.permute(0, 2, 1).clone())))
    with torch.no_grad(), _EphemeralQueue(eph) as q:
        y = m(x, t, length, xf_proj=xf_proj, xf_out=xf_out)
        assert q.i == len(eph), (q.i, len(eph))
    for h in hooks:
        h.remove()
    out = {"x": x, "timesteps": t, "length": length, "xf_proj": xf_proj, "xf_out": xf_out, "output": y}
    for k, v in trace.items():
        if trace_level == "all" or k.endswith(("dual_self_attn", "cross_attn", "ffn", "sd_cross_attn", "top2_idx", "gap23")) \
                or k in ("fused_emb",):
            out["trace/" + k] = v
    for nm, buf in m.named_buffers():
        if nm.endswith(("expert_usage", "expert_importance")):
            out["buf/" + nm] = buf.clone()
    meta = dict(cfg=cfg, B=B, T=T, N=N, wseed=wseed, iseed=iseed, latent_dim=D, text_latent_dim=Dt)
    save(name, out, meta)
    print(f"{name}: out absmax {y.abs().max():.4f}  keys {len(out)}")


def save(name, tensors, meta):
    arrs = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in tensors.items()}
    arrs["__meta__"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrs)


class _NoiseQueue:
    def __init__(self, noises):
        self.noises = noises
        self.i = 0
        self.orig = torch.randn_like

    def __enter__(self):
        q = self

        def rl(x, *a, **k):
            n = q.noises[q.i]
            q.i += 1
            assert n.shape == x.shape
            return n.clone()

        torch.randn_like = rl
        return self

    def __exit__(self, *a):
        torch.randn_like = self.orig


def case_loops(name, cfg, B, T, N, wseed, iseed, steps_cfg, steps_ddim):
    m, D = build_reference(cfg, wseed)
    Dt = cfg["text_latent_dim_arg"]
    eph = synth.synth_ephemerals(D, Dt, cfg["num_layers"], wseed)
    x, _, length, xf_proj, xf_out = synth.synth_inputs(B, T, cfg["input_feats"], N, Dt, iseed, min_len=4)
    xo_u = synth.uniform_pm1((1, N, Dt), "in.uncond", iseed).expand(B, N, Dt).contiguous() * (3.0 ** 0.5)
    xp_u = xo_u.mean(dim=1)
    _StubTextEncoder.table = {"uncond": (xp_u, xo_u), "cond": (xf_proj, xf_out)}
    out = {"x_T": x, "length": length, "xf_proj": xf_proj, "xf_out": xf_out, "xf_proj_uncond": xp_u,
           "xf_out_uncond": xo_u}
    kw = {"xf_proj": xf_proj, "xf_out": xf_out, "length": length, "text": ["a person walks"] * B}
    # --- CFG DDPM (the trainer's path, ddpm_trainer.py:161-173: clip_denoised=False) -------------------
    diff = RG.GaussianDiffusion(betas=RG.get_named_beta_schedule("linear", steps_cfg),
                                model_mean_type=RG.ModelMeanType.EPSILON,
                                model_var_type=RG.ModelVarType.FIXED_SMALL, loss_type=RG.LossType.MSE)
    noises = [synth.uniform_pm1((B, T, cfg["input_feats"]), f"noise.cfg.{i}", iseed) * (3.0 ** 0.5) for i in range(steps_cfg)]
    traj = []
    orig_step = diff.p_sample_with_cfg

    def rec(*a, **k):
        o = orig_step(*a, **k)
        traj.append(o["sample"].clone())
        return o

    diff.p_sample_with_cfg = rec
    with _EphemeralQueue(eph), _NoiseQueue(noises):
        y = diff.p_sample_loop_with_cfg(m, (B, T, cfg["input_feats"]), noise=x.clone(), clip_denoised=False,
                                        model_kwargs=kw, cfg_scale=2.5)
    out["cfg/final"] = y
    out["cfg/traj_idx"] = torch.tensor([0, steps_cfg // 2, steps_cfg - 1])
    out["cfg/traj"] = torch.stack([traj[i] for i in (0, steps_cfg // 2, steps_cfg - 1)])
    # step noises are regenerated in tests from synth.uniform_pm1(f"noise.cfg.{i}", iseed) * sqrt(3)
    # --- DDIM (gaussian_diffusion.py:744-818), eta 0 and 0.5, default clip_denoised=True ---------------
    diff2 = RG.GaussianDiffusion(betas=RG.get_named_beta_schedule("linear", steps_ddim),
                                 model_mean_type=RG.ModelMeanType.EPSILON,
                                 model_var_type=RG.ModelVarType.FIXED_SMALL, loss_type=RG.LossType.MSE)
    kw2 = {"xf_proj": xf_proj, "xf_out": xf_out, "length": length}
    for eta in (0.0, 0.5):
        noises = [synth.uniform_pm1((B, T, cfg["input_feats"]), f"noise.ddim.{eta}.{i}", iseed) * (3.0 ** 0.5)
                  for i in range(steps_ddim)]
        with _EphemeralQueue(eph), _NoiseQueue(noises):
            y = diff2.ddim_sample_loop(m, (B, T, cfg["input_feats"]), noise=x.clone(), model_kwargs=kw2, eta=eta)
        out[f"ddim{eta}/final"] = y
    # tables pinned too
    for nm in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_recip_alphas_cumprod",
               "sqrt_recipm1_alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped",
               "posterior_mean_coef1", "posterior_mean_coef2"):
        out["tables/" + nm] = getattr(diff, nm)
    d1000 = RG.GaussianDiffusion(betas=RG.get_named_beta_schedule("linear", 1000),
                                 model_mean_type=RG.ModelMeanType.EPSILON,
                                 model_var_type=RG.ModelVarType.FIXED_SMALL, loss_type=RG.LossType.MSE)
    for nm in ("posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2",
               "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod"):
        out["tables1000/" + nm] = getattr(d1000, nm)
    meta = dict(cfg=cfg, B=B, T=T, N=N, wseed=wseed, iseed=iseed, latent_dim=D, text_latent_dim=Dt,
                steps_cfg=steps_cfg, steps_ddim=steps_ddim, cfg_scale=2.5)
    save(name, out, meta)
    print(f"{name}: cfg final absmax {out['cfg/final'].abs().max():.4f}")


def caption_embedding(caption: str, N: int, Dt: int, seed: int):
    """Stub text embedding of ONE caption: N token rows (what a text encoder would return for it), a function of the string."""
    return synth.uniform_pm1((N, Dt), "cap." + caption, seed) * (3.0 ** 0.5)


def case_trainer_generate(name, cfg, captions, m_lens, batch_size, N_cond, N_uncond, wseed, iseed, steps, cfg_scale):
    """The reference's own DDPMTrainer.generate (trainers/ddpm_trainer.py:145-199) on the tiny model: captions -> stub text
    embeddings (N_cond tokens per caption, N_uncond for the empty caption: a real tokenizer pads "" to fewer tokens than the
    captions, and the uncond forward re-encodes [""] * B every step, gaussian_diffusion.py:1059-1062), m_lens, queued noise
    -> the list generate() returns.  Pins the whole trainer call: T = min(m_lens.max(), num_frames), batching by
    batch_size, clip_denoised=False, cfg_scale from args, x_T = th.randn(*shape) and randn_like per step."""
    import types
    with contextlib.redirect_stdout(io.StringIO()):
        import trainers.ddpm_trainer as RTr
    m, D = build_reference(cfg, wseed)
    Dt, Fe = cfg["text_latent_dim_arg"], cfg["input_feats"]
    eph = synth.synth_ephemerals(D, Dt, cfg["num_layers"], wseed)
    xo_u1 = caption_embedding("", N_uncond, Dt, iseed)

    class _Enc(nn.Module):
        def forward(self, text, device):
            if all(t == "" for t in text):
                xo = xo_u1[None].expand(len(text), -1, -1).contiguous()
            else:
                xo = torch.stack([caption_embedding(t, N_cond, Dt, iseed) for t in text])
            return xo.mean(dim=1), xo

    m.text_encoder = _Enc()
    args = types.SimpleNamespace(device=torch.device("cpu"), diffusion_steps=steps, is_train=False, cfg_scale=cfg_scale)
    tr = RTr.DDPMTrainer(args, m)
    m_lens_t = torch.tensor(m_lens)
    # noise queues in the order generate() consumes them: per batch, x_T then one randn_like per step
    nb = (len(captions) + batch_size - 1) // batch_size
    xT, step_noise = [], []
    for k in range(nb):
        lo, hi = k * batch_size, min((k + 1) * batch_size, len(captions))
        T = min(int(m_lens_t[lo:hi].max()), cfg["num_frames"])
        xT.append(synth.uniform_pm1((hi - lo, T, Fe), f"gen.xT.{k}", iseed) * (3.0 ** 0.5))
        step_noise.append([synth.uniform_pm1((hi - lo, T, Fe), f"gen.noise.{k}.{i}", iseed) * (3.0 ** 0.5) for i in range(steps)])
    flat = [n for k in range(nb) for n in step_noise[k]]
    orig_randn = torch.randn
    qi = [0]

    def randn(*shape, **kw):
        x = xT[qi[0]]
        qi[0] += 1
        assert tuple(shape) == tuple(x.shape), (shape, x.shape)
        return x.clone()

    traj = []
    orig_step = tr.diffusion.p_sample_with_cfg

    def rec(*a, **k):
        o = orig_step(*a, **k)
        traj.append(o["sample"].clone())
        return o

    tr.diffusion.p_sample_with_cfg = rec
    torch.randn = randn
    try:
        with _EphemeralQueue(eph), _NoiseQueue(flat), contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            outs = tr.generate(list(captions), m_lens_t, Fe, batch_size=batch_size)
    finally:
        torch.randn = orig_randn
    assert len(outs) == len(captions) and qi[0] == nb
    out = {"m_lens": m_lens_t}
    for i, o in enumerate(outs):
        out[f"out/{i}"] = o
    out["traj0/idx"] = torch.tensor([0, steps // 2, steps - 1])
    out["traj0"] = torch.stack([traj[i] for i in (0, steps // 2, steps - 1)])  # first batch's x_t after those steps
    meta = dict(cfg=cfg, captions=list(captions), batch_size=batch_size, N_cond=N_cond, N_uncond=N_uncond, wseed=wseed,
                iseed=iseed, steps=steps, cfg_scale=cfg_scale, latent_dim=D, text_latent_dim=Dt)
    save(name, out, meta)
    print(f"{name}: {len(outs)} samples, shapes {[tuple(o.shape) for o in outs]}, absmax {max(float(o.abs().max()) for o in outs):.4f}")


def case_layout():
    lay = {}
    for tag, kw in {
        "small_E8_L4": dict(latent_dim=512, ff_size=1024, num_layers=4, num_heads=4, text_latent_dim=256,
                            moe_num_experts=8, model_size="small", num_frames=196),
        "big_E8_L1": dict(latent_dim=512, ff_size=1024, num_layers=1, num_heads=4, text_latent_dim=256,
                          moe_num_experts=8, model_size="big", num_frames=196),
        "tools_L2": dict(latent_dim=512, ff_size=256, num_layers=2, num_heads=4, text_latent_dim=128,
                         moe_num_experts=4, model_size="small", num_frames=196),
    }.items():
        with contextlib.redirect_stdout(io.StringIO()):
            m = RT.MotionTransformer(263, **kw)
        lay[tag] = {"kwargs": kw, "keys": [[k, list(v.shape)] for k, v in m.state_dict().items()],
                    "n_params": sum(p.numel() for p in m.parameters())}
        del m
    with open(os.path.join(OUT, "state_dict_layout.json"), "w") as f:
        json.dump(lay, f)
    print("layout:", {k: len(v["keys"]) for k, v in lay.items()})


def case_projection_qr():
    import models.fast_attention as RF
    out = {}
    for dh in (16, 128):
        with contextlib.redirect_stdout(io.StringIO()):
            fa = RF.FastAttention(dim=4 * dh, head_dim=dh, num_features=256)
        torch.manual_seed(5)
        out[f"P{dh}"] = fa._create_projection(torch.device("cpu"))
    save("projection_qr", out, {"seed": 5})
    print("projection_qr:", {k: tuple(v.shape) for k, v in out.items()})


def case_text_head():
    """Runs the reference's EnhancedTextEncoder.forward (text_encoder.py:24-43) with the two network fetches replaced:
    AutoModel / AutoTokenizer.from_pretrained return tiny deterministic stand-ins, so the reference's own prompt
    concatenation, projection head and pooling produce the golden outputs."""
    import types
    import models.text_encoder as TE

    Hs, Dt, B, N0 = 96, 32, 3, 5
    hidden = synth.uniform_pm1((B, N0, Hs), "text_head.hidden", 31) * (3.0 ** 0.5)

    class _Bert(nn.Module):
        config = types.SimpleNamespace(hidden_size=Hs)

        def forward(self, input_ids=None, attention_mask=None, return_dict=True):
            return types.SimpleNamespace(last_hidden_state=hidden.clone())

    class _Tok:
        def __call__(self, text, **kw):
            ids = torch.zeros(len(text), N0, dtype=torch.long)
            return _Batch(ids)

    class _Batch:
        def __init__(self, ids):
            self.input_ids, self.attention_mask = ids, torch.ones_like(ids)

        def to(self, device):
            return self

    real_m, real_t = TE.AutoModel.from_pretrained, TE.AutoTokenizer.from_pretrained
    TE.AutoModel.from_pretrained = staticmethod(lambda name: _Bert())
    TE.AutoTokenizer.from_pretrained = staticmethod(lambda name: _Tok())
    try:
        enc = TE.EnhancedTextEncoder(output_dim=Dt).eval()
    finally:
        TE.AutoModel.from_pretrained, TE.AutoTokenizer.from_pretrained = real_m, real_t
    sd = {"proj.0.weight": 1.0 + 0.1 * synth.uniform_pm1((Hs,), "text_head.ln_w", 31),
          "proj.0.bias": 0.1 * synth.uniform_pm1((Hs,), "text_head.ln_b", 31),
          "proj.1.weight": synth.uniform_pm1((Dt, Hs), "text_head.w", 31) * Hs ** -0.5,
          "proj.1.bias": 0.1 * synth.uniform_pm1((Dt,), "text_head.b", 31),
          "prompt_tokens": synth.uniform_pm1((1, 8, Hs), "text_head.prompts", 31) * (3.0 ** 0.5)}
    missing = enc.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys and all(k.startswith("bert.") for k in missing.missing_keys), missing
    with torch.no_grad():
        pooled, projected = enc(["a person walks"] * B, torch.device("cpu"))
    out = {"hidden": hidden, "pooled": pooled, "projected": projected}
    out.update({"sd/" + k: v for k, v in sd.items()})
    save("text_head", out, {"Hs": Hs, "Dt": Dt, "B": B, "N0": N0, "P": 8, "seed": 31})
    print("text_head:", tuple(pooled.shape), tuple(projected.shape))


def case_motion_post():
    """recover_from_ric (utils/motion_process.py:403-416) and motion_temporal_filter (utils/utils.py:125-130) of the
    reference on seeded motions, driven exactly as tools/visualization.py:21-27,89 does."""
    import utils.motion_process as MP
    import utils.utils as UU
    B, T, Fe = 3, 24, 263
    motion = synth.uniform_pm1((B, T, Fe), "motion_post.x", 41) * (3.0 ** 0.5)
    mean = synth.uniform_pm1((Fe,), "motion_post.mean", 41).numpy() * 0.2
    std = (0.5 + 0.25 * synth.uniform_pm1((Fe,), "motion_post.std", 41)).numpy()
    lengths = [24, 17, 8]
    out = {"motion": motion, "mean": torch.from_numpy(mean), "std": torch.from_numpy(std),
           "length": torch.tensor(lengths, dtype=torch.int64)}
    for b, n in enumerate(lengths):
        data = motion[b, :n].numpy() * std + mean                                   # visualization.py:89
        joint = MP.recover_from_ric(torch.from_numpy(data).float(), 22).numpy()      # :22
        out[f"joints_raw/{b}"] = torch.from_numpy(joint.copy())
        out[f"joints/{b}"] = torch.from_numpy(UU.motion_temporal_filter(joint, sigma=1))  # :24
    save("motion_post", out, {"B": B, "T": T, "feats": Fe, "joints": 22, "sigma": 1, "seed": 41})
    print("motion_post:", {k: tuple(v.shape) for k, v in out.items() if k.startswith("joints/")})


def case_moe_loss():
    """SwitchMoELayer.get_load_balancing_loss (switch_moe.py:113-145) of the reference on given counters, and
    MotionTransformer.get_moe_loss (transformer.py:272-279) summed over a model's layers."""
    out = {}
    for E in (4, 8):
        with contextlib.redirect_stdout(io.StringIO()):
            layer = RS.SwitchMoELayer(16, 32, num_experts=E)
        usage = (synth.uniform_pm1((E,), f"moe_loss.usage{E}", 51).abs() * 100).round()
        imp = synth.uniform_pm1((E,), f"moe_loss.imp{E}", 51).abs() * 37.0
        layer.expert_usage.copy_(usage), layer.expert_importance.copy_(imp)
        out[f"usage{E}"], out[f"importance{E}"] = usage, imp
        out[f"loss{E}"] = layer.get_load_balancing_loss().detach().reshape(1)
    layer.expert_usage.zero_(), layer.expert_importance.zero_()
    out["loss_zero_counters"] = layer.get_load_balancing_loss().detach().reshape(1)
    save("moe_loss", out, {"seed": 51})
    print("moe_loss:", {k: float(v) for k, v in out.items() if k.startswith("loss")})


def cfgd(D, F_, H, Dt, E, L, size="small", frames=196, feats=263, **extra):
    d = dict(input_feats=feats, num_frames=frames, latent_dim_arg=D, ff_size_arg=F_, num_heads=H,
             text_latent_dim_arg=Dt, moe_num_experts=E, num_layers=L, model_size=size)
    d.update(extra)
    return d


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    if "--text-head-only" in sys.argv:
        case_text_head()
        return
    if "--motion-post-only" in sys.argv:
        case_motion_post()
        return
    if "--moe-loss-only" in sys.argv:
        case_moe_loss()
        return
    if "--trainer-only" in sys.argv:
        case_trainer_generate("trainer_generate", cfgd(64, 128, 4, 32, 4, 1, frames=16), ["a person walks", "someone jumps", "a man sits down"],
                              [16, 12, 10], 2, 6, 4, wseed=18, iseed=28, steps=25, cfg_scale=2.5)
        return
    if "--loops-only" not in sys.argv:
        case_text_head()
        case_motion_post()
        case_moe_loss()
        case_layout()
        case_projection_qr()
        case_forward("fwd_tiny", cfgd(64, 128, 4, 32, 4, 1, frames=16), B=2, T=16, N=6, wseed=11, iseed=21)
        case_forward("fwd_tiny_l2", cfgd(64, 128, 4, 32, 4, 2, frames=24, odd_length=True), B=3, T=12, N=5, wseed=12, iseed=22)
        case_forward("fwd_tiny_eqdim", cfgd(64, 96, 2, 64, 3, 1, frames=16), B=2, T=8, N=4, wseed=13, iseed=23)
        case_forward("fwd_small_dims", cfgd(512, 1024, 4, 256, 8, 1), B=2, T=16, N=6, wseed=14, iseed=24, trace_level="main")
        case_forward("fwd_big_dims", cfgd(512, 1024, 4, 256, 8, 1, size="big"), B=2, T=8, N=5, wseed=15, iseed=25, trace_level="main")
        case_forward("fwd_tools_shape", cfgd(512, 256, 4, 128, 4, 2), B=2, T=12, N=7, wseed=16, iseed=26, trace_level="main")
    case_loops("loops_tiny", cfgd(64, 128, 4, 32, 4, 1, frames=16), B=2, T=16, N=6, wseed=17, iseed=27, steps_cfg=25, steps_ddim=25)
    case_trainer_generate("trainer_generate", cfgd(64, 128, 4, 32, 4, 1, frames=16), ["a person walks", "someone jumps", "a man sits down"],
                          [16, 12, 10], 2, 6, 4, wseed=18, iseed=28, steps=25, cfg_scale=2.5)


if __name__ == "__main__":
    main()
