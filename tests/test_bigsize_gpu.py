"""GPU: BASELINE.json configs[2] / configs[3] at their own sizes -- the BIG model (latent 1024, ff 2048, text 512, head_dim
256), 8 experts, L = 4, B = 32, T = 196.  As for configs[1] the oracle cannot run the whole batch in seconds, so parity is
shown on single samples of the full-batch HIP run (samples never interact) plus one guided CFG step
(cond + uncond rows batched as 2B = 64) and teacher-forced DDIM steps of a 100-step schedule (configs[3]'s sampler)."""
import os
import sys

import pytest
import torch

from conftest import ROOT, pkg, rel_inf

sys.path.insert(0, os.path.join(ROOT, "oracle"))
import denoiser_ref as R  # noqa: E402
import diffusion_ref as DR  # noqa: E402

from test_round2_gpu import RouteDump, count_flips, _layer_names  # noqa: E402

pytestmark = pytest.mark.gpu

_CACHE = {}


def _big(precision, E=8, seed=0):
    """The big model with synthetic weights (built once per precision per session: 1.08 G parameters)."""
    key = (precision, E, seed)
    if key not in _CACHE:
        T_, synth = pkg("transformer"), pkg("synth")
        m = T_.MotionTransformer(263, num_frames=196, latent_dim=512, ff_size=1024, num_layers=4, num_heads=4,
                                 text_latent_dim=256, moe_num_experts=E, model_size="big", precision=precision)
        if "host" not in _CACHE:
            sd = synth.synth_state_dict(m._layout, seed)
            eph = synth.synth_ephemerals(1024, 512, 4, 7)
            proj = synth.synth_projections(256, 4, 7)
            _CACHE["host"] = dict(sd=sd, eph_list=eph, proj_list=proj, eph={n: (w, b) for n, w, b in eph}, proj=dict(proj),
                                  mcfg=dict(latent_dim=1024, num_heads=4, num_layers=4, moe_num_experts=E))
        host = _CACHE["host"]
        m.load_state_dict(host["sd"], strict=True)
        m.set_ephemerals(host["eph_list"]), m.set_projections(host["proj_list"])
        _CACHE[key] = m.cuda().eval()
    return _CACHE[key], _CACHE["host"]


def _inputs(B=32, T=196):
    synth = pkg("synth")
    x, _, length, xf_proj, xf_out = synth.synth_inputs(B, T, 263, 28, 512, 0, min_len=40)
    return x, length, xf_proj, xf_out


def _oracle(host, x, t, length, xf_proj, xf_out, trace=None):
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    with torch.no_grad():
        return R.denoiser_forward(host["sd"], host["mcfg"], x, t, length, xf_proj, xf_out, host["eph"], host["proj"], None, trace)


# precision -> (max rel-inf error, max fraction of routing decisions that differ, max median per-frame error); see
# tests/test_round2_gpu.py::MODE_BUDGET for the reasoning.  The big widths run the generic (GEMM-composed) attention path.
BIG_BUDGET = {3: (1e-3, 0.0, 1e-4), 2: (None, 3e-2, 8e-3)}


@pytest.mark.parametrize("precision", [3, 2])
def test_configs2_one_sample_of_the_full_batch_matches_the_oracle(precision):
    B, T, L = 32, 196, 4
    m, host = _big(precision)
    x, length, xf_proj, xf_out = _inputs(B, T)
    t = torch.full((B,), 977, dtype=torch.int64)
    with RouteDump(2 * L, B, T) as rd:
        y = m(x.cuda(), t.cuda(), length.cuda(), xf_proj=xf_proj.cuda(), xf_out=xf_out.cuda()).cpu()
    assert y.shape == (B, T, 263) and torch.isfinite(y).all()
    tol, flip_frac, med_tol = BIG_BUDGET[precision]
    b = 17
    trace = {}
    sl = slice(b, b + 1)
    ref = _oracle(host, x[sl], t[sl], length[sl], xf_proj[sl], xf_out[sl], trace)
    flips = decisions = 0
    for li, name in enumerate(_layer_names(L)):
        S = T // 2 if li < L else T
        ours = rd.layer(li, B * S).reshape(2, B, S, 2)[:, b]
        want = torch.stack([trace[f"{name}.ffn.branches.{br}.top2_idx"] for br in range(2)]).reshape(2, S, 2)
        flips += count_flips(ours, want)[0]
        decisions += 2 * S
    err = rel_inf(y[sl], ref)
    frame = (y[b] - ref[0]).abs().amax(-1) / ref.abs().max()
    print(f"big precision {precision} sample {b} (length {int(length[b])}): rel err {err:.2e}, median frame err "
          f"{float(frame.median()):.2e}, routing decisions that differ: {flips}/{decisions}")
    assert flips <= flip_frac * decisions
    assert float(frame.median()) < med_tol
    if tol is not None:
        assert err < tol


def test_configs2_guided_step_on_the_full_batch():
    """configs[2]: CFG 7.5 with cond + uncond rows batched as 2B = 64 rows through ONE forward + the fused posterior update;
    sample 3 of the step's output against the oracle's two forwards + cfg_step for that sample; and the batched forward must
    equal two separate B-row forwards bit for bit (what the captured sampling step relies on)."""
    B, T, steps, scale, t0 = 32, 196, 1000, 7.5, 811
    m, host = _big(3)
    D, synth = pkg("diffusion"), pkg("synth")
    x, length, xf_proj, xf_out = _inputs(B, T)
    xo_u = synth.uniform_pm1((1, 28, 512), "in.uncond", 0) * (3.0 ** 0.5)
    xp_u = xo_u.mean(1)
    m.set_uncond_embedding(xp_u.cuda(), xo_u.cuda())
    diff = D.GaussianDiffusion(betas=D.get_named_beta_schedule("linear", steps), model_mean_type=D.ModelMeanType.EPSILON,
                               model_var_type=D.ModelVarType.FIXED_SMALL, loss_type=D.LossType.MSE)
    noise = synth.uniform_pm1((B, T, 263), "noise.big", 0) * (3.0 ** 0.5)
    tt = torch.full((B,), t0, dtype=torch.int64)
    kw = {"xf_proj": xf_proj.cuda(), "xf_out": xf_out.cuda(), "length": length.cuda(), "text": ["x"] * B}
    out = diff.p_sample_with_cfg(m, x.cuda(), tt.cuda(), clip_denoised=False, model_kwargs=kw, cfg_scale=scale, noise=noise.cuda())
    b = 3
    sl = slice(b, b + 1)
    ec = _oracle(host, x[sl], tt[sl], length[sl], xf_proj[sl], xf_out[sl])
    eu = _oracle(host, x[sl], tt[sl], length[sl], xp_u, xo_u)
    ref, ref0 = DR.cfg_step(DR.Tables(DR.linear_betas(steps)), t0, x[sl], ec, eu, noise[sl], scale)
    e1, e0 = rel_inf(out["sample"][sl].cpu(), ref), rel_inf(out["pred_xstart"][sl].cpu(), ref0)
    print(f"big guided step, sample {b}: x_(t-1) {e1:.2e}, pred_xstart {e0:.2e}")
    assert e1 < 1e-3 and e0 < 1e-3
    xd, ld, td = x.cuda(), length.cuda(), tt.cuda()
    xu_p, xu_o = m.uncond_embedding(B, "cuda")
    both = m(torch.cat([xd, xd]), torch.cat([td, td]), torch.cat([ld, ld]), xf_proj=torch.cat([kw["xf_proj"], xu_p]),
             xf_out=torch.cat([kw["xf_out"], xu_o]))
    assert torch.equal(both[:B], m(xd, td, ld, xf_proj=kw["xf_proj"], xf_out=kw["xf_out"]))
    assert torch.equal(both[B:], m(xd, td, ld, xf_proj=xu_p, xf_out=xu_o))


def test_configs3_ddim_steps_of_a_100_step_schedule():
    """configs[3]'s sampler: DDIM on a diffusion built with 100 betas (the reference has no respacing, SURVEY.md section 0
    fact 8), big model, B = 4 ragged samples at T = 64.  The oracle runs the loop; every 9th step of the HIP sampler is
    checked from the oracle's state (eta 0 and 0.5), then the captured 100-step loop must be finite and deterministic."""
    B, T, steps = 4, 64, 100
    m, host = _big(3)
    D, synth = pkg("diffusion"), pkg("synth")
    x, _, length, xf_proj, xf_out = synth.synth_inputs(B, T, 263, 28, 512, 5, min_len=24)
    diff = D.GaussianDiffusion(betas=D.get_named_beta_schedule("linear", steps), model_mean_type=D.ModelMeanType.EPSILON,
                               model_var_type=D.ModelVarType.FIXED_SMALL, loss_type=D.LossType.MSE)
    kw = {"xf_proj": xf_proj.cuda(), "xf_out": xf_out.cuda(), "length": length.cuda()}
    tb = DR.Tables(DR.linear_betas(steps))
    xs = x.clone()
    worst, clean, flipped = 0.0, 0, 0
    L = 4
    names = _layer_names(L)
    for i, t in enumerate(reversed(range(steps))):
        if i % 9 and t:
            continue  # the oracle advances on the visited steps only: each check starts from an oracle state anyway
        tt = torch.full((B,), t, dtype=torch.int64)
        trace = {}
        eps = _oracle(host, xs, tt, length, xf_proj, xf_out, trace)
        for eta in (0.0, 0.5):
            nz = synth.uniform_pm1((B, T, 263), f"noise.ddim100.{i}", 5) * (3.0 ** 0.5)
            ref, ref0 = DR.ddim_step(tb, t, xs, eps, nz, eta, clip=False)
            with RouteDump(2 * L, B, T) as rd:
                out = diff.ddim_sample(m, xs.cuda(), tt.cuda(), clip_denoised=False, model_kwargs=kw, eta=eta, noise=nz.cuda())
            flips = 0
            for li, name in enumerate(names):
                S = T // 2 if li < L else T
                want = torch.stack([trace[f"{name}.ffn.branches.{br}.top2_idx"] for br in range(2)])
                flips += count_flips(rd.layer(li, B * S), want)[0]
            err = max(rel_inf(out["sample"].cpu(), ref), rel_inf(out["pred_xstart"].cpu(), ref0))
            if flips == 0:  # a flipped near-tie is an O(1) local change of a different (equally valid) forward: not gated
                worst, clean = max(worst, err), clean + 1
            else:
                flipped += 1
                print(f"  step t={t} eta={eta}: {flips} routing decisions differ from the oracle's, error {err:.2e} (not gated)")
        xs = ref
    print(f"100-step DDIM, big model: worst teacher-forced step error {worst:.2e} over {clean} flip-free checks ({flipped} with flips)")
    assert clean >= 12 and worst < 1e-3
    a = diff.ddim_sample_loop(m, (B, T, 263), clip_denoised=False, model_kwargs=kw, eta=0.0, seed=5)
    b = diff.ddim_sample_loop(m, (B, T, 263), clip_denoised=False, model_kwargs=kw, eta=0.0, seed=5)
    assert torch.isfinite(a).all() and torch.equal(a, b)


@pytest.mark.parametrize("precision,tol", [(2, 6e-4), (1, 4e-3)])
@pytest.mark.parametrize("S", [40, 98, 196, 17, 224])  # 17: one full + one ragged tile; 224: the largest S the kernels take
def test_big_width_performer_core_fused_kernel(S, precision, tol):
    """head_dim 256 (big model): mdm_performer_attn_forward runs the fused core of csrc/perf_attn2.hip (feature / value halves,
    KV^T through an L2-resident scratch) in the 16-bit modes -- against the oracle's PerformerSelfAttention, both attention
    slots, ragged lengths, S not a multiple of the tile sizes; knob 23 selects the GEMM-composed chain on the same inputs."""
    import ctypes as C
    import torch.nn.functional as F
    from conftest import build_module, load_golden
    g, meta = load_golden("fwd_big_dims")
    m, (sd, eph, proj, mcfg) = build_module(meta, precision=precision)
    L, synth = pkg("_lib"), pkg("synth")
    lib, pm = L.lib(), m.pack()
    D, H, B = 1024, 4, 4
    h = synth.uniform_pm1((B, S, D), "blk.h", S) * 1.5
    emb = synth.uniform_pm1((B, D), "blk.emb", S)
    length = torch.tensor([S, max(1, S - 13), max(1, S // 2), 0])  # the last sample has every key masked
    pre = "decoder_blocks_low.0.module"
    mask = R.src_mask(S, length)
    ws = m._workspace(B, S, 1)
    hd, ld = h.cuda().contiguous(), length.to(torch.int32).cuda()
    for which, slot in ((0, "local"), (1, "global")):
        sp = f"{pre}.dual_self_attn.{slot}_attn.style_block"
        w, b = eph[f"low.0.{slot}_style"]
        sc = F.linear(F.silu(F.linear(emb, w, b)), sd[sp + ".emb_layers.1.weight"], sd[sp + ".emb_layers.1.bias"]).cuda().contiguous()

        def run():
            out = torch.empty_like(hd)
            L.check(lib.mdm_performer_attn_forward(C.byref(pm.model), C.c_int32(0), C.c_int32(which), C.c_void_p(hd.data_ptr()),
                                                   C.c_void_p(sc.data_ptr()), C.c_void_p(ld.data_ptr()), C.c_int32(B), C.c_int32(S),
                                                   C.c_void_p(out.data_ptr()), C.c_void_p(ws.data_ptr()), C.c_int64(ws.numel()),
                                                   C.c_int32(precision), C.c_void_p(L.stream_ptr())))
            return out.cpu()

        fused = run()
        lib.mdm_set_gemm_variant(23)
        try:
            chain = run()
        finally:
            lib.mdm_set_gemm_variant(0)
        with torch.no_grad():
            ref = R.performer_self_attention(h, emb, mask, sd, f"{pre}.dual_self_attn.{slot}_attn", H, eph[f"low.0.{slot}_style"],
                                             proj[f"low.0.{slot}"])
        e_f, e_c = rel_inf(fused, ref), rel_inf(chain, ref)
        print(f"big performer {slot} S={S} precision {precision}: fused {e_f:.2e}, GEMM-composed chain {e_c:.2e}")
        assert e_f < tol and not torch.equal(fused, chain)


@pytest.mark.parametrize("precision,tol", [(2, 6e-3), (1, 3e-2)])
@pytest.mark.parametrize("S,N", [(196, 28), (98, 9), (37, 85), (5, 96), (1, 1)])  # 96: the largest N of the softmax core
def test_big_width_linear_cross_attention_fused_core(S, N, precision, tol):
    """head_dim 256: GatedCrossAttention (fast_attention.py:242-272) through mdm_block_forward runs the fused
    softmax_dh(q) A core of csrc/xattn.hip (lin_xattn256_kernel: A^T of one (batch, head) resident in 132 KiB of LDS) in the
    16-bit modes; knob 23 selects the GEMM-composed path (head softmax + batched GEMM) on the same inputs."""
    import ctypes as C
    import torch.nn.functional as F
    from conftest import build_module, load_golden
    g, meta = load_golden("fwd_big_dims")
    m, (sd, eph, proj, mcfg) = build_module(meta, precision=precision)
    L, synth = pkg("_lib"), pkg("synth")
    lib, pm = L.lib(), m.pack()
    D, H, B = 1024, 4, 2
    Dt = sd["decoder_blocks_low.0.module.cross_attn.base_ca.key.weight"].shape[1]
    h = synth.uniform_pm1((B, S, D), "blk.h", S) * 1.5
    emb = synth.uniform_pm1((B, D), "blk.emb", S)
    xf = synth.uniform_pm1((B, N, Dt), "blk.xf", N) * 1.7
    length = torch.tensor([S, max(1, S - 13)])
    pre = "decoder_blocks_low.0.module"
    sc = []
    for slot, sp in (("local_style", pre + ".dual_self_attn.local_attn.style_block"),
                     ("global_style", pre + ".dual_self_attn.global_attn.style_block"),
                     ("cross_style", pre + ".cross_attn.base_ca.proj_out"), ("ffn_style", pre + ".ffn.proj_out")):
        w, b = eph["low.0." + slot]
        sc.append(F.linear(F.silu(F.linear(emb, w, b)), sd[sp + ".emb_layers.1.weight"], sd[sp + ".emb_layers.1.bias"]))
    sc = torch.stack(sc)
    tcache = m.prepare_text(xf.cuda())
    ws = m._workspace(B, S, N)
    hd, scd, ld = h.cuda().contiguous(), sc.cuda().contiguous(), length.to(torch.int32).cuda()

    def run():
        out = torch.empty_like(hd)
        L.check(lib.mdm_block_forward(C.byref(pm.model), C.c_int32(0), C.c_int32(L.BLOCK_CROSS), C.byref(tcache["tc"]),
                                      C.c_void_p(hd.data_ptr()), C.c_void_p(scd.data_ptr()), C.c_void_p(ld.data_ptr()),
                                      C.c_int32(B), C.c_int32(S), C.c_void_p(out.data_ptr()), C.c_void_p(ws.data_ptr()),
                                      C.c_int64(ws.numel()), C.c_void_p(0), C.c_int32(precision), C.c_void_p(L.stream_ptr())))
        return out.cpu()

    fused = run()
    lib.mdm_set_gemm_variant(23)
    try:
        chain = run()
    finally:
        lib.mdm_set_gemm_variant(0)
    with torch.no_grad():
        ref = R.gated_cross_attention(h, xf, emb, sd, pre + ".cross_attn", H, eph["low.0.cross_style"])
    e_f, e_c = rel_inf(fused, ref), rel_inf(chain, ref)
    print(f"big linear cross-attention S={S} N={N} precision {precision}: fused {e_f:.2e}, GEMM-composed {e_c:.2e}")
    assert e_f < tol and not torch.equal(fused, chain)

    # MemoryEfficientCrossAttentionBlock (fast_attention.py:301-330) on the same inputs: the fused scores / softmax / PV core
    # of csrc/xattn.hip at head_dim 256 (sd_attn_kernel<.., 256>) against the oracle and the GEMM-composed path
    def run_sd():
        out = torch.empty_like(hd)
        L.check(lib.mdm_block_forward(C.byref(pm.model), C.c_int32(0), C.c_int32(L.BLOCK_SDCROSS), C.byref(tcache["tc"]),
                                      C.c_void_p(hd.data_ptr()), C.c_void_p(scd.data_ptr()), C.c_void_p(ld.data_ptr()),
                                      C.c_int32(B), C.c_int32(S), C.c_void_p(out.data_ptr()), C.c_void_p(ws.data_ptr()),
                                      C.c_int64(ws.numel()), C.c_void_p(0), C.c_int32(precision), C.c_void_p(L.stream_ptr())))
        return out.cpu()

    fused = run_sd()
    lib.mdm_set_gemm_variant(23)
    try:
        chain = run_sd()
    finally:
        lib.mdm_set_gemm_variant(0)
    with torch.no_grad():
        ref = R.softmax_cross_ffn(h, xf, sd, pre + ".sd_cross_attn", H)
    e_f, e_c = rel_inf(fused, ref), rel_inf(chain, ref)
    print(f"big softmax cross-attention + FFN S={S} N={N} precision {precision}: fused core {e_f:.2e}, GEMM-composed {e_c:.2e}")
    assert e_f < tol
    # in the bf16 mode the GEMM-composed path rounds the same operands to bf16 and accumulates in the same MFMA order: the two
    # paths may agree bit for bit; in the fp16 mode the composed path still rounds to bf16 and must differ
    assert precision == 1 or not torch.equal(fused, chain)


@pytest.mark.parametrize("precision", [2, 1])
def test_big_width_router_with_compile_time_expert_count_is_bit_identical(precision):
    """D = 1024, E = 8: the router instantiated for the expert count and hn format (knob 26) against the run-time one
    (knob 27) through a whole forward at a ragged batch -- same arithmetic in the same order, so bit-equal outputs."""
    B, T = 3, 38   # 3 x 19 and 3 x 38 tokens: neither a multiple of the 16 tokens a workgroup iteration takes
    m, host = _big(precision)
    x, length, xf_proj, xf_out = _inputs(B, T)
    t = torch.full((B,), 500, dtype=torch.int64)
    lib = pkg("_lib").lib()
    outs = {}
    for knob in (27, 26):
        lib.mdm_set_gemm_variant(knob)
        try:
            outs[knob] = m(x.cuda(), t.cuda(), length.cuda(), xf_proj=xf_proj.cuda(), xf_out=xf_out.cuda()).cpu()
        finally:
            lib.mdm_set_gemm_variant(0)
    default = m(x.cuda(), t.cuda(), length.cuda(), xf_proj=xf_proj.cuda(), xf_out=xf_out.cuda()).cpu()
    assert torch.isfinite(outs[26]).all()
    assert torch.equal(outs[26], outs[27]), float((outs[26] - outs[27]).abs().max())
    assert torch.equal(default, outs[27])


@pytest.mark.parametrize("precision", [2, 1])
def test_big_width_streamed_weight_linears_compute_the_tile_kernels_function(precision):
    """csrc/gemm_stream.hip inside the model (MdmPacked.ws: the per-layer D x D Linears of the 16-bit modes at D = 1024) against the
    same forward on the tile kernel (knob 63).  Same MFMA, operand order and k order; with general epilogue scales the two differ
    in the last fp32 bit (7e-8 relative, tests/test_gemm_gpu.py), which a 16-bit store may round either way: gated at 16-bit noise."""
    L_ = pkg("_lib")
    m, _ = _big(precision)
    assert m.pack().wstream1, "the big model packs fragment streams for its D x D Linears"
    B, T = 4, 64
    x, length, xf_proj, xf_out = _inputs(B, T)
    length = length.clamp(max=T)
    t = torch.full((B,), 500, dtype=torch.int64)
    args = (x.cuda(), t.cuda(), length.cuda())
    kw = dict(xf_proj=xf_proj.cuda(), xf_out=xf_out.cuda())
    y = m(*args, **kw).cpu()
    L_.lib().mdm_set_gemm_variant(63)
    try:
        yt = m(*args, **kw).cpu()
    finally:
        L_.lib().mdm_set_gemm_variant(0)
    frame = (y - yt).abs().amax(-1) / yt.abs().max()
    print(f"big precision {precision}: streamed vs tile Linears, median frame difference {float(frame.median()):.2e}, max {float(frame.max()):.2e}")
    assert torch.isfinite(y).all()  # (with the model's epilogues -- alpha = r1_scale = 1 -- the two kernels may agree bit for bit)
    assert float(frame.median()) < (2e-3 if precision == 2 else 1.5e-2)
