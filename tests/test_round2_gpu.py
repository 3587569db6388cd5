"""GPU: precision modes (bf16 / fp16 / bf16x3 / mixed) with measured error AND routing-flip counts, the hardening fixes
(non-finite rows, CFG halves with different token counts), counter-based noise, and the parity pins that round 1 only
checked against itself (unguided ancestral step, unclipped DDIM loop, DDPMTrainer.generate, direct C-ABI block calls)."""
import ctypes as C
import os
import sys
import types

import pytest
import torch
import torch.nn.functional as F

from conftest import ROOT, build_module, load_golden, pkg, rel_inf

sys.path.insert(0, os.path.join(ROOT, "oracle"))
import denoiser_ref as R  # noqa: E402
import diffusion_ref as DR  # noqa: E402
import philox_ref as P  # noqa: E402

pytestmark = pytest.mark.gpu


def _layer_names(L):
    return [f"decoder_blocks_{s}.{i}.module" for s in ("low", "high") for i in range(L)]


class RouteDump:
    """mdm_route_dump as a context manager: .idx[layer, branch, token, k] of the LAST forward inside the block."""

    def __init__(self, L2, B, T):
        self.buf = torch.full((L2, 2, B * T, 2), -1, dtype=torch.int32, device="cuda")

    def __enter__(self):
        pkg("_lib").lib().mdm_route_dump(C.c_void_p(self.buf.data_ptr()), C.c_int64(self.buf.numel()))
        return self

    def __exit__(self, *a):
        torch.cuda.synchronize()
        pkg("_lib").lib().mdm_route_dump(C.c_void_p(0), C.c_int64(0))

    def layer(self, li, M):
        """(2, M, 2) decisions of layer li, whose token count is M (the buffer is strided for the full scale)."""
        flat = self.buf[li].reshape(-1)[:4 * M]
        return flat.reshape(2, M, 2).cpu()


def count_flips(ours, ref):
    """tokens whose top-2 SET differs, and tokens whose ordered pair differs (ours / ref: (.., 2) int)."""
    a, b = ours.long().sort(-1).values, ref.long().sort(-1).values
    return int((a != b).any(-1).sum()), int((ours.long() != ref.long()).any(-1).sum())


# ---------------------------------------------------------------------------------------------------------------------
# precision modes on the reference-generated goldens
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", ["fwd_tiny", "fwd_small_dims", "fwd_big_dims", "fwd_tools_shape"])
@pytest.mark.parametrize("precision,tol_forced,tol_free", [(2, 6e-3, 6e-3), (4, 1e-3, 1e-3)])
def test_f16_and_mixed_modes_against_reference_goldens(case, precision, tol_forced, tol_free):
    g, meta = load_golden(case)
    m, _ = build_module(meta, precision=precision)
    B, T, _ = g["x"].shape
    L = meta["cfg"]["num_layers"]
    forced = torch.zeros((2 * L, 2 * 2 * B * T), dtype=torch.int32)
    refidx = []
    for li, name in enumerate(_layer_names(L)):
        idx = torch.stack([g[f"trace/{name}.ffn.branches.{b}.moe.top2_idx"] for b in range(2)])  # (2, M, 2)
        forced[li, :idx.numel()] = idx.reshape(-1).to(torch.int32)
        refidx.append(idx)
    args = (g["x"].cuda(), g["timesteps"].cuda(), g["length"].cuda())
    kw = dict(xf_proj=g["xf_proj"].cuda(), xf_out=g["xf_out"].cuda())
    e_forced = rel_inf(m(*args, forced_routing=forced, **kw).cpu(), g["output"])
    with RouteDump(2 * L, B, T) as rd:
        y = m(*args, **kw).cpu()
    e_free = rel_inf(y, g["output"])
    flips = sum(count_flips(rd.layer(li, refidx[li].shape[1]), refidx[li])[0] for li in range(2 * L))
    total = sum(2 * r.shape[1] for r in refidx)
    print(f"{case} precision {precision}: rel err {e_forced:.2e} (reference routing), {e_free:.2e} free routing, "
          f"{flips}/{total} routing decisions differ")
    assert e_forced < tol_forced
    # fwd_tiny holds one near-tie (p2 - p3 ~ 1e-4) that every reduced-precision mode resolves the other way: at most that one
    assert flips <= (1 if case == "fwd_tiny" else 0)
    if flips == 0:
        assert e_free < tol_free


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE configs[1] size (small, E=8, L=4, B=32, T=196), FREE routing: error bound + flip budget per mode
# ---------------------------------------------------------------------------------------------------------------------
def _build_full(E, precision, seed=0):
    T_ = pkg("transformer")
    synth = pkg("synth")
    m = T_.MotionTransformer(263, num_frames=196, latent_dim=512, ff_size=1024, num_layers=4, num_heads=4,
                             text_latent_dim=256, moe_num_experts=E, model_size="small", precision=precision)
    sd = synth.synth_state_dict(m._layout, seed)
    m.load_state_dict(sd, strict=True)
    eph = synth.synth_ephemerals(512, 256, 4, 7)
    proj = synth.synth_projections(128, 4, 7)
    m.set_ephemerals(eph), m.set_projections(proj)
    host = dict(sd=sd, eph={n: (w, b) for n, w, b in eph}, proj=dict(proj),
                mcfg=dict(latent_dim=512, num_heads=4, num_layers=4, moe_num_experts=E))
    return m.cuda().eval(), host


# precision -> (max rel-inf error with free routing, max fraction of routing decisions that may differ from the oracle's,
#  max median per-frame error).  The fp32-grade mode holds the north-star 1e-3 with ZERO flips.  Every reduced-precision
#  mode flips some near-ties (1.4 % of tokens have p2 - p3 < 1e-3): a flipped token is an O(1) local
#  change, so the max error of those modes is a flip artefact and is reported, not gated; what is gated is the flip
#  fraction and the median per-frame error (the arithmetic error).  Measured on MI355X (samples 0 / 17 vs the oracle; whole
#  batch vs the fp32-grade run in tools/mode_compare.py): mixed 0.09 - 0.9 % flips (0.23 % over the batch), median 1.1e-3;
#  fp16 1.3 - 2.4 % (0.8 %), 2.0e-3; bf16 3.8 - 5.1 %, 1.5e-2 - 2.0e-2.  The ragged sample (17) flips more than the full one.
MODE_BUDGET = {3: (1e-3, 0.0, 1e-4), 4: (None, 1.5e-2, 2.5e-3), 2: (None, 4e-2, 5e-3), 1: (None, 1e-1, 4e-2)}
_ORACLE_CACHE = {}


def _oracle_sample(host, x, t, length, xf_proj, xf_out, b):
    if b not in _ORACLE_CACHE:
        torch.set_num_threads(min(16, os.cpu_count() or 1))
        sl = slice(b, b + 1)
        trace = {}
        with torch.no_grad():
            ref = R.denoiser_forward(host["sd"], host["mcfg"], x[sl], t[sl], length[sl], xf_proj[sl], xf_out[sl],
                                     host["eph"], host["proj"], None, trace)
        _ORACLE_CACHE[b] = (ref, trace)
    return _ORACLE_CACHE[b]


@pytest.mark.parametrize("precision", [3, 4, 2, 1])
def test_configs1_free_routing_error_and_flip_budget(precision):
    """The WHOLE B=32 batch goes through the HIP forward with free routing; samples 0 (full length) and 17 (ragged) are
    compared with the oracle run on that sample alone (samples never interact), and every one of their 2 x 8 x 1.5 T
    routing decisions is compared with the oracle's."""
    B, T, L = 32, 196, 4
    synth = pkg("synth")
    m, host = _build_full(8, precision)
    x, _, length, xf_proj, xf_out = synth.synth_inputs(B, T, 263, 28, 256, 0, min_len=40)
    t = torch.full((B,), 977, dtype=torch.int64)
    with RouteDump(2 * L, B, T) as rd:
        y = m(x.cuda(), t.cuda(), length.cuda(), xf_proj=xf_proj.cuda(), xf_out=xf_out.cuda()).cpu()
    tol, flip_frac, med_tol = MODE_BUDGET[precision]
    names = _layer_names(L)
    for b in (0, 17):
        ref, trace = _oracle_sample(host, x, t, length, xf_proj, xf_out, b)
        flips = decisions = 0
        for li, name in enumerate(names):
            S = T // 2 if li < L else T
            ours = rd.layer(li, B * S).reshape(2, B, S, 2)[:, b]                        # (2, S, 2)
            want = torch.stack([trace[f"{name}.ffn.branches.{br}.top2_idx"] for br in range(2)]).reshape(2, S, 2)
            flips += count_flips(ours, want)[0]
            decisions += 2 * S
        err = rel_inf(y[b:b + 1], ref)
        frame = (y[b] - ref[0]).abs().amax(-1) / ref.abs().max()
        print(f"precision {precision} sample {b} (length {int(length[b])}): rel err {err:.2e}, median frame err "
              f"{float(frame.median()):.2e}, frames off by > 5 %: {int((frame > 0.05).sum())}/{T}, "
              f"routing decisions that differ from the oracle's: {flips}/{decisions}")
        assert flips <= flip_frac * decisions, (precision, b, flips, decisions)
        assert float(frame.median()) < med_tol, (precision, b, float(frame.median()))
        if tol is not None:
            assert err < tol, (precision, b, err)


def test_configs0_free_running_loop_drift_is_attributed_to_routing_flips():
    """BASELINE configs[0] (small, E=4, B=2, T=64, 50-step CFG DDPM), fp32-grade mode, free-running HIP loop vs the
    free-running oracle loop: per step, the state difference AND the number of routing decisions (cond + uncond
    forwards) that differ from the oracle's.  The trajectories must agree to 1e-3 for as long as no decision has
    flipped; after the first flip they are two different (equally valid) samples."""
    B, T, steps, scale, L = 2, 64, 50, 7.5, 4
    synth, D = pkg("synth"), pkg("diffusion")
    m, host = _build_full(4, 3, seed=3)
    x, _, length, xf_proj, xf_out = synth.synth_inputs(B, T, 263, 28, 256, 3, min_len=40)
    xo_u = synth.uniform_pm1((1, 28, 256), "in.uncond", 3) * (3.0 ** 0.5)
    xp_u = xo_u.mean(1)
    m.set_uncond_embedding(xp_u.cuda(), xo_u.cuda())
    diff = D.GaussianDiffusion(betas=D.get_named_beta_schedule("linear", steps), model_mean_type=D.ModelMeanType.EPSILON,
                               model_var_type=D.ModelVarType.FIXED_SMALL, loss_type=D.LossType.MSE)
    noises = [synth.uniform_pm1((B, T, 263), f"noise.c0.{i}", 3) * (3.0 ** 0.5) for i in range(steps)]
    kw = {"xf_proj": xf_proj.cuda(), "xf_out": xf_out.cuda(), "length": length.cuda(), "text": ["x"] * B}
    names = _layer_names(L)
    ours_route, traj = [], []
    with RouteDump(2 * L, 2 * B, T) as rd:
        def cb(i, t, xx):
            traj.append(xx.clone().cpu())
            ours_route.append(rd.buf.clone().cpu())
        diff.p_sample_loop_with_cfg(m, (B, T, 263), noise=x.cuda(), clip_denoised=False, model_kwargs=kw, cfg_scale=scale,
                                    step_noise=noises, callback=cb, use_graph=False)
    tb = DR.Tables(DR.linear_betas(steps))
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    xs = x.clone()
    first_flip, errs, flips_per_step = None, [], []
    for i in range(steps):
        t = steps - 1 - i
        tt = torch.full((B,), t, dtype=torch.int64)
        tr_c, tr_u = {}, {}
        with torch.no_grad():
            ec = R.denoiser_forward(host["sd"], host["mcfg"], xs, tt, length, xf_proj, xf_out, host["eph"], host["proj"], None, tr_c)
            eu = R.denoiser_forward(host["sd"], host["mcfg"], xs, tt, length, xp_u.expand(B, -1), xo_u.expand(B, -1, -1),
                                    host["eph"], host["proj"], None, tr_u)
        xs, _ = DR.cfg_step(tb, t, xs, ec, eu, noises[i], scale)
        flips = 0
        for li, name in enumerate(names):
            S = T // 2 if li < L else T
            ours = ours_route[i][li].reshape(-1)[:4 * 2 * B * S].reshape(2, 2 * B, S, 2)   # rows: cond samples, uncond samples
            for br in range(2):
                want = torch.cat([tr_c[f"{name}.ffn.branches.{br}.top2_idx"].reshape(B, S, 2),
                                  tr_u[f"{name}.ffn.branches.{br}.top2_idx"].reshape(B, S, 2)])
                flips += count_flips(ours[br], want)[0]
        flips_per_step.append(flips)
        errs.append(rel_inf(traj[i], xs))
        if flips and first_flip is None:
            first_flip = i
        if first_flip is not None and i >= first_flip + 3:
            break  # past the first flip the two loops are different samples: nothing left to attribute
    print("per-step routing flips:", flips_per_step, " state rel err:", [f"{e:.1e}" for e in errs], " first flip at step", first_flip)
    clean = len(errs) if first_flip is None else first_flip
    assert clean >= 2, "the fp32-grade loop flipped a routing decision within its first two steps"
    assert all(e < 1e-3 for e in errs[:clean]), errs[:clean]


# ---------------------------------------------------------------------------------------------------------------------
# hardening
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("precision", [3, 1])
def test_non_finite_row_gives_nan_output_not_a_fault(precision):
    """A NaN / Inf activation row must come out as NaN for THAT sample (router indices stay in range: no LDS or global
    out-of-bounds access) and leave the other samples bit-identical."""
    g, meta = load_golden("fwd_small_dims")
    m, _ = build_module(meta, precision=precision)
    B, T, _ = g["x"].shape
    args = (g["timesteps"].cuda(), g["length"].cuda())
    kw = dict(xf_proj=g["xf_proj"].cuda(), xf_out=g["xf_out"].cuda())
    clean = m(g["x"].cuda(), *args, **kw)
    for poison in (float("nan"), float("inf")):
        x = g["x"].clone()
        x[0, 3, 7] = poison
        with RouteDump(2, B, T) as rd:
            y = m(x.cuda(), *args, **kw)
        torch.cuda.synchronize()
        assert not torch.isfinite(y[0]).all()
        assert torch.equal(y[1:], clean[1:])
        for li in range(2):
            idx = rd.layer(li, B * (T // 2 if li == 0 else T))
            assert int(idx.min()) >= 0 and int(idx.max()) < meta["cfg"]["moe_num_experts"]
            assert bool((idx[..., 0] != idx[..., 1]).all())
    with pytest.raises(ValueError):
        m(g["x"].cuda(), *args, forced_routing=torch.full((2, 4 * B * T), 99, dtype=torch.int32), **kw)


@pytest.mark.parametrize("ragged_text", ["mask", "split"])
def test_cfg_halves_with_different_token_counts_match_the_oracle(ragged_text):
    """The empty caption tokenises to fewer tokens than the captions.  Default ("mask"): the shorter half is padded
    and the text cache carries per-row token counts -- still one forward of 2B rows; "split": cond and uncond as two forwards
    with their own text caches.  Either must equal the oracle's two-forward CFG step."""
    g, meta = load_golden("fwd_small_dims")
    m, (sd, eph, proj, mcfg) = build_module(meta, precision=3)
    m.ragged_text = ragged_text
    D = pkg("diffusion")
    synth = pkg("synth")
    B, T, Fe = g["x"].shape
    Dt = meta["text_latent_dim"]
    xo_u = (synth.uniform_pm1((1, 3, Dt), "u.short", 5) * 1.7).expand(B, -1, -1).contiguous()   # 3 tokens vs the captions' 6
    xp_u = xo_u.mean(1)
    assert xo_u.shape[1] != g["xf_out"].shape[1]
    m.set_uncond_embedding(xp_u[:1].cuda(), xo_u[:1].cuda())
    steps, scale, t = 50, 2.5, 31
    diff = D.GaussianDiffusion(betas=D.get_named_beta_schedule("linear", steps), model_mean_type=D.ModelMeanType.EPSILON,
                               model_var_type=D.ModelVarType.FIXED_SMALL, loss_type=D.LossType.MSE)
    noise = synth.uniform_pm1((B, T, Fe), "n.split", 5)
    tt = torch.full((B,), t, dtype=torch.int64)
    kw = {"xf_proj": g["xf_proj"].cuda(), "xf_out": g["xf_out"].cuda(), "length": g["length"].cuda(), "text": ["x"] * B}
    out = diff.p_sample_with_cfg(m, g["x"].cuda(), tt.cuda(), clip_denoised=False, model_kwargs=kw, cfg_scale=scale, noise=noise.cuda())
    with torch.no_grad():
        ec = R.denoiser_forward(sd, mcfg, g["x"], tt, g["length"], g["xf_proj"], g["xf_out"], eph, proj)
        eu = R.denoiser_forward(sd, mcfg, g["x"], tt, g["length"], xp_u, xo_u, eph, proj)
    ref, ref0 = DR.cfg_step(DR.Tables(DR.linear_betas(steps)), t, g["x"], ec, eu, noise, scale)
    assert rel_inf(out["sample"].cpu(), ref) < 1e-3 and rel_inf(out["pred_xstart"].cpu(), ref0) < 1e-3
    # and through the captured loop (graph replay of two forwards per step)
    y = diff.p_sample_loop_with_cfg(m, (B, T, Fe), noise=g["x"].cuda(), clip_denoised=False, model_kwargs=kw, cfg_scale=scale, seed=11)
    assert torch.isfinite(y).all()
    if ragged_text == "mask":   # the two ways of running ragged halves give the same trajectory
        m.ragged_text = "split"
        y2 = diff.p_sample_loop_with_cfg(m, (B, T, Fe), noise=g["x"].cuda(), clip_denoised=False, model_kwargs=kw, cfg_scale=scale, seed=11)
        assert rel_inf(y.cpu(), y2.cpu()) < 1e-3


def test_wrong_device_inputs_raise():
    g, meta = load_golden("fwd_tiny")
    m, _ = build_module(meta, precision=3)
    L = pkg("_lib")
    with pytest.raises(L.MdmError):
        m(g["x"], g["timesteps"], g["length"], xf_proj=g["xf_proj"], xf_out=g["xf_out"])
    d = pkg("diffusion")
    diff = d.GaussianDiffusion(betas=d.get_named_beta_schedule("linear", 50), model_mean_type=d.ModelMeanType.EPSILON,
                               model_var_type=d.ModelVarType.FIXED_SMALL, loss_type=d.LossType.MSE)
    kw = {"xf_proj": g["xf_proj"].cuda(), "xf_out": g["xf_out"].cuda(), "length": g["length"].cuda()}
    with pytest.raises(ValueError):
        diff.ddim_sample(m, g["x"].cuda(), torch.tensor([50, 50]), model_kwargs=kw)   # t outside the schedule


# ---------------------------------------------------------------------------------------------------------------------
# counter-based noise: device generator vs its numpy restatement, and shard invariance of the REAL sampler
# ---------------------------------------------------------------------------------------------------------------------
def test_philox_noise_matches_the_oracle_and_is_shard_invariant():
    L = pkg("_lib")
    per, n, first, seed = 16 * 263, 5, 7, 0x1234_5678_9ABC
    out = torch.empty((n, per), device="cuda")
    for stream_imm in (3, L.NOISE_STREAM_XT):
        L.check(L.lib().mdm_noise_normal(C.c_void_p(out.data_ptr()), C.c_int64(per), C.c_int32(n), C.c_int64(first),
                                         C.c_uint64(seed), C.c_void_p(0), C.c_int32(stream_imm), C.c_void_p(L.stream_ptr())))
        ref = torch.from_numpy(P.normal(per, n, first, seed, stream_imm))
        assert float((out.cpu() - ref).abs().max()) < 2e-6
    assert abs(float(out.mean())) < 0.02 and abs(float(out.std()) - 1.0) < 0.02
    t_dev = torch.tensor([3], dtype=torch.int32, device="cuda")   # the captured step reads its timestep from the device
    part = torch.empty((2, per), device="cuda")
    L.check(L.lib().mdm_noise_normal(C.c_void_p(part.data_ptr()), C.c_int64(per), C.c_int32(2), C.c_int64(first + 2),
                                     C.c_uint64(seed), C.c_void_p(t_dev.data_ptr()), C.c_int32(0), C.c_void_p(L.stream_ptr())))
    want = torch.from_numpy(P.normal(per, n, first, seed, 3))[2:4]
    assert float((part.cpu() - want).abs().max()) < 2e-6


@pytest.mark.parametrize("mode", ["cfg", "ddim"])
def test_real_sampler_on_two_half_batch_shards_equals_the_unsharded_run(mode):
    """SURVEY.md 8(e) on one GPU: the REAL sampler (captured graph, device noise) on rows [0, 2) and [2, 4) of a batch,
    run one after the other with sample_offset, must equal the 4-row run bit for bit."""
    g, meta = load_golden("loops_tiny")
    m, _ = build_module(meta, precision=3)
    D, synth, dmod = pkg("diffusion"), pkg("synth"), pkg("dist")
    B, T, Fe, N, Dt = 4, 16, g["x_T"].shape[2], 6, meta["text_latent_dim"]
    _, _, length, xf_proj, xf_out = synth.synth_inputs(B, T, Fe, N, Dt, 9, min_len=4)
    m.set_uncond_embedding(g["xf_proj_uncond"][:1].cuda(), g["xf_out_uncond"][:1].cuda())
    diff = D.GaussianDiffusion(betas=D.get_named_beta_schedule("linear", 25), model_mean_type=D.ModelMeanType.EPSILON,
                               model_var_type=D.ModelVarType.FIXED_SMALL, loss_type=D.LossType.MSE)
    kw = {"xf_proj": xf_proj.cuda(), "xf_out": xf_out.cuda(), "length": length.cuda(), "text": ["x"] * B}

    def run(shape, kw_local, seed, first):
        if mode == "cfg":
            return diff.p_sample_loop_with_cfg(m, shape, clip_denoised=False, model_kwargs=kw_local, cfg_scale=2.5, seed=seed,
                                               sample_offset=first)
        return diff.ddim_sample_loop(m, shape, clip_denoised=False, model_kwargs=kw_local, eta=0.5, seed=seed, sample_offset=first)

    whole = run((B, T, Fe), kw, 77, 0)
    parts = []
    for rank in range(2):
        lo, hi = dmod.shard_range(B, rank, 2)
        parts.append(run((hi - lo, T, Fe), dmod.shard_kwargs(kw, lo, hi), 77, lo))
    assert torch.equal(torch.cat(parts), whole)
    assert not torch.equal(whole, run((B, T, Fe), kw, 78, 0))


# ---------------------------------------------------------------------------------------------------------------------
# parity pins that round 1 only compared with themselves
# ---------------------------------------------------------------------------------------------------------------------
def _loops_setup():
    g, meta = load_golden("loops_tiny")
    m, host = build_module(meta, precision=3)
    D, synth = pkg("diffusion"), pkg("synth")
    B, T, Fe = g["x_T"].shape
    kw = {"xf_proj": g["xf_proj"].cuda(), "xf_out": g["xf_out"].cuda(), "length": g["length"].cuda()}

    def diff(steps):
        return D.GaussianDiffusion(betas=D.get_named_beta_schedule("linear", steps), model_mean_type=D.ModelMeanType.EPSILON,
                                   model_var_type=D.ModelVarType.FIXED_SMALL, loss_type=D.LossType.MSE)

    def noises(tag, n):
        return [synth.uniform_pm1((B, T, Fe), f"noise.{tag}.{i}", meta["iseed"]) * (3.0 ** 0.5) for i in range(n)]

    sd, eph, proj, mcfg = host

    def oracle_model(x, t, cond=True):
        with torch.no_grad():
            return R.denoiser_forward(sd, mcfg, x, t, g["length"], g["xf_proj"], g["xf_out"], eph, proj)

    return g, meta, m, diff, noises, kw, oracle_model


def test_unguided_ancestral_step_and_loop_match_the_oracle():
    """p_sample / p_sample_loop (gaussian_diffusion.py:582-646 with the intended noise draw; the reference raises at :606)
    against the oracle's cfg_step(eps_u=None): single steps at several t, then the whole 25-step loop."""
    g, meta, m, diff, noises, kw, oracle_model = _loops_setup()
    steps = 25
    d, tb = diff(steps), DR.Tables(DR.linear_betas(steps))
    x = g["x_T"]
    B = x.shape[0]
    nz = noises("p", steps)
    for t in (steps - 1, steps // 2, 1, 0):
        tt = torch.full((B,), t, dtype=torch.int64)
        for clip in (False, True):
            out = d.p_sample(m, x.cuda(), tt.cuda(), clip_denoised=clip, model_kwargs=kw, noise=nz[0].cuda())
            ref, ref0 = DR.cfg_step(tb, t, x, oracle_model(x, tt), None, nz[0], 0.0, clip=clip)
            assert rel_inf(out["sample"].cpu(), ref) < 1e-3 and rel_inf(out["pred_xstart"].cpu(), ref0) < 1e-3, (t, clip)
    y = d.p_sample_loop(m, tuple(x.shape), noise=x.cuda(), clip_denoised=False, model_kwargs=kw, step_noise=nz)
    xs = x.clone()
    for i, t in enumerate(reversed(range(steps))):
        tt = torch.full((B,), t, dtype=torch.int64)
        xs, _ = DR.cfg_step(tb, t, xs, oracle_model(xs, tt), None, nz[i], 0.0, clip=False)
    assert rel_inf(y.cpu(), xs) < 1e-3


@pytest.mark.parametrize("eta", [0.0, 0.5])
def test_unclipped_ddim_loop_matches_the_oracle_at_1e3(eta):
    """ddim_sample_loop with clip_denoised=False against the oracle's loop: the step arithmetic and the forward hold 1e-3
    over the whole 25-step loop.  (The reference's default clip_denoised=True is checked against the reference golden in
    test_sampler_gpu.py with a looser bound: an element of pred_xstart that sits within rounding of the +-1 clamp is clamped
    in one implementation and not in the other, and eps is then re-derived from the clamped value with a ~1e2 gain.)"""
    g, meta, m, diff, noises, kw, oracle_model = _loops_setup()
    steps = meta["steps_ddim"]
    d, tb = diff(steps), DR.Tables(DR.linear_betas(steps))
    nz = noises(f"ddim.{eta}", steps)
    y = d.ddim_sample_loop(m, tuple(g["x_T"].shape), noise=g["x_T"].cuda(), clip_denoised=False, model_kwargs=kw, eta=eta, step_noise=nz)
    ref = DR.ddim_loop(lambda x, t, c: oracle_model(x, t), tb, g["x_T"], nz, eta=eta, clip_denoised=False)
    err = rel_inf(y.cpu(), ref)
    print(f"unclipped DDIM eta={eta}: {err:.2e}")
    assert err < 1e-3


def test_trainer_generate_matches_the_reference_trainer_golden():
    """DDPMTrainer.generate (trainers/ddpm_trainer.py:145-199) end to end against a golden produced by the REFERENCE's
    own trainer: captions -> stub embeddings (6 tokens per caption, 4 for the empty caption), m_lens, batch_size 2,
    T = min(m_lens.max(), num_frames) per batch, cfg_scale from args, queued x_T / step noise."""
    g, meta = load_golden("trainer_generate")
    m, _ = build_module(meta, precision=3)
    synth, Tr = pkg("synth"), pkg("trainer")
    Dt, Fe, steps, bs = meta["text_latent_dim"], meta["cfg"]["input_feats"], meta["steps"], meta["batch_size"]

    def cap_emb(c, N):
        return synth.uniform_pm1((N, Dt), "cap." + c, meta["iseed"]) * (3.0 ** 0.5)

    def enc(text, device):
        if all(t == "" for t in text):
            xo = cap_emb("", meta["N_uncond"])[None].expand(len(text), -1, -1).contiguous()
        else:
            xo = torch.stack([cap_emb(t, meta["N_cond"]) for t in text])
        return xo.mean(1).to(device), xo.to(device)

    m.text_encoder_fn = enc
    args = types.SimpleNamespace(device=torch.device("cuda"), diffusion_steps=steps, is_train=False, cfg_scale=meta["cfg_scale"])
    tr = Tr.DDPMTrainer(args, m)
    caps, lens = meta["captions"], g["m_lens"]
    noises = []
    for k in range((len(caps) + bs - 1) // bs):
        lo, hi = k * bs, min((k + 1) * bs, len(caps))
        T = min(int(lens[lo:hi].max()), meta["cfg"]["num_frames"])
        noises.append((synth.uniform_pm1((hi - lo, T, Fe), f"gen.xT.{k}", meta["iseed"]) * (3.0 ** 0.5),
                       [synth.uniform_pm1((hi - lo, T, Fe), f"gen.noise.{k}.{i}", meta["iseed"]) * (3.0 ** 0.5) for i in range(steps)]))
    outs = tr.generate(caps, lens, Fe, batch_size=bs, noises=noises)
    assert len(outs) == len(caps)
    for i, o in enumerate(outs):
        assert o.shape == g[f"out/{i}"].shape
        err = rel_inf(o.cpu(), g[f"out/{i}"])
        print(f"generate sample {i}: {err:.2e}")
        assert err < 1e-3, (i, err)


def test_stylization_and_stem_entry_points_match_the_oracle():
    """mdm_stylization_forward and mdm_stem_embeddings called directly through the C ABI (round 1 only checked that the
    symbols exist): StylizationBlock.forward (stylization.py:20-31) and the fused time/text embedding + the per-block
    (scale | shift) rows (transformer.py:313-321, stylization.py:22-27) against the oracle."""
    g, meta = load_golden("fwd_small_dims")
    m, (sd, eph, proj, mcfg) = build_module(meta, precision=3)
    L, synth = pkg("_lib"), pkg("synth")
    lib, pm = L.lib(), m.pack()
    D, Dt, Ln = 512, 256, meta["cfg"]["num_layers"]
    B, S = 3, 50
    h = synth.uniform_pm1((B, S, D), "sty.h", 1) * 1.5
    emb = synth.uniform_pm1((B, D), "sty.emb", 1)
    pre = "decoder_blocks_low.0.module.cross_attn.base_ca.proj_out"
    with torch.no_grad():
        ref = R.stylization(h, emb, sd, pre, eph["low.0.cross_style"])
        w, b = eph["low.0.cross_style"]
        sc = F.linear(F.silu(F.linear(emb, w, b)), sd[pre + ".emb_layers.1.weight"], sd[pre + ".emb_layers.1.bias"])
    hd, scd = h.cuda(), sc.cuda().contiguous()
    tmp, out = torch.empty_like(hd), torch.empty_like(hd)
    st = pm.layers[0].ca_style
    L.check(lib.mdm_stylization_forward(C.byref(st), C.c_void_p(hd.data_ptr()), C.c_void_p(scd.data_ptr()), C.c_int32(B),
                                        C.c_int32(S), C.c_int32(D), C.c_void_p(tmp.data_ptr()), C.c_void_p(out.data_ptr()),
                                        C.c_int32(3), C.c_void_p(L.stream_ptr())))
    assert rel_inf(out.cpu(), ref) < 1e-3
    # stem
    ts = torch.tensor([0, 501, 999])
    xp = synth.uniform_pm1((B, Dt), "stem.xp", 2)
    ts_d, xp_d = ts.cuda(), xp.cuda().contiguous()  # keep the device copies alive across the call
    emb_out = torch.empty((B, D), device="cuda")
    sc_out = torch.empty((8 * Ln, B, 2 * D), device="cuda")
    ws = m._workspace(B, 2, 1)
    L.check(lib.mdm_stem_embeddings(C.byref(pm.model), C.c_void_p(ts_d.data_ptr()), C.c_void_p(xp_d.data_ptr()),
                                    C.c_int32(B), C.c_void_p(emb_out.data_ptr()), C.c_void_p(sc_out.data_ptr()),
                                    C.c_void_p(ws.data_ptr()), C.c_int64(ws.numel()), C.c_int32(3), C.c_void_p(L.stream_ptr())))
    with torch.no_grad():
        emb_ref = R.fused_embedding(sd, ts, xp, D, eph)
    assert rel_inf(emb_out.cpu(), emb_ref) < 1e-3
    slots = [("local_style", "dual_self_attn.local_attn.style_block"), ("global_style", "dual_self_attn.global_attn.style_block"),
             ("cross_style", "cross_attn.base_ca.proj_out"), ("ffn_style", "ffn.proj_out")]
    for li, name in enumerate(_layer_names(Ln)):
        tag = name.split("_")[2].replace(".module", "")  # "low.0"
        for si, (slot, sub) in enumerate(slots):
            w, b = eph[f"{tag}.{slot}"]
            with torch.no_grad():
                want = F.linear(F.silu(F.linear(emb_ref, w, b)), sd[f"{name}.{sub}.emb_layers.1.weight"], sd[f"{name}.{sub}.emb_layers.1.bias"])
            assert rel_inf(sc_out[4 * li + si].cpu(), want) < 1e-3, (name, slot)
