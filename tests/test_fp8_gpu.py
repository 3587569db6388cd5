"""GPU: fp8 (e4m3) expert GEMMs -- BASELINE configs[4] ("big, 16 experts, top-2, fp8 MFMA expert GEMMs").
  * the fp8 GEMM kernel (csrc/gemm8.hip, block-scaled MFMA K = 128) against an fp64 product of the SAME quantised operands
    (so the check is exact up to accumulation order), dense / grouped / gathered / ragged;
  * the library's quantiser against torch's float8_e4m3fn rounding;
  * the denoiser in the fp8 mode at the big widths with E = 16 against the oracle: error and routing flips REPORTED and
    gated on a budget ("bf16/fp8 runs report their error and routing-flip count")."""
import os
import sys

import pytest
import torch

from conftest import ROOT, pkg, rel_inf

sys.path.insert(0, os.path.join(ROOT, "oracle"))
import denoiser_ref as R  # noqa: E402

from test_round2_gpu import RouteDump, count_flips, _layer_names  # noqa: E402

pytestmark = pytest.mark.gpu
FP8_FORCED_MEDIAN_BUDGET = 0.15  # measured 8.8e-2 at L = 4, T = 196 (4.0e-2 at L = 2, T = 64): the e4m3 error grows with the depth


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return ((torch.rand(*shape, generator=g) * 2 - 1) * scale).cuda()


def _deq(q8, scale, K):
    return q8.view(torch.float8_e4m3fn)[:, :K].double() * scale.double()[:, None]


def test_quantiser_matches_torch_e4m3():
    ops = pkg("ops")
    x = _rand(37, 300, seed=1, scale=3.0)
    x[5] = 0
    q8, sc = ops.quantize_rows_fp8(x)
    amax = x.abs().amax(1)
    want_sc = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax))
    assert torch.allclose(sc, want_sc, rtol=1e-6)
    want = (x / sc[:, None]).to(torch.float8_e4m3fn).view(torch.uint8)
    assert torch.equal(q8[:, :300], want) and int(q8[:, 300:].max()) == 0 and q8.shape[1] == 384


@pytest.mark.parametrize("M,K,N", [(300, 512, 1024), (77, 1024, 512), (1000, 2048, 1024), (1, 128, 128)])
def test_dense_fp8_gemm(M, K, N):
    L, ops = pkg("_lib"), pkg("ops")
    a, w, b = _rand(M, K, seed=2, scale=2.0), _rand(N, K, seed=3, scale=K ** -0.5), _rand(N, seed=4, scale=0.1)
    a8, asc = ops.quantize_rows_fp8(a)
    pw = ops.PackedWeight(w, fmt="f8")
    y = ops.gemm_fp8(a8, asc, pw, b, act=L.ACT_GELU)
    ref = torch.nn.functional.gelu(_deq(a8, asc, K) @ _deq(pw.hi, pw.lo, K).T + b.double())
    assert rel_inf(y.cpu(), ref.float().cpu()) < 2e-5
    true = torch.nn.functional.gelu(a.double() @ w.double().T + b.double())
    print(f"fp8 GEMM {M}x{K}x{N}: exact on quantised operands; vs the fp32 operands {rel_inf(y.cpu(), true.float().cpu()):.2e}")


def test_fp8_hidden_output_saturates_instead_of_turning_nan():
    """e4m3fn has no infinity, so a hidden activation above 448 / 8 = 56 would convert to NaN and poison the row
    through the second GEMM.  The e4m3 output of the kernel saturates at +-448."""
    L, ops = pkg("_lib"), pkg("ops")
    M, K, N = 64, 128, 128
    a = _rand(M, K, seed=5, scale=1.0)
    a[3] *= 400.0   # a row of outliers: GELU(x) ~ x up to several hundred -> 8 x that is far beyond 448
    w, b = _rand(N, K, seed=6, scale=1.0), _rand(N, seed=7, scale=0.1)
    a8, asc = ops.quantize_rows_fp8(a)
    pw = ops.PackedWeight(w, fmt="f8")
    hid8 = torch.zeros((M, N), dtype=torch.uint8, device="cuda")
    ops.gemm_fp8(a8, asc, pw, b, act=L.ACT_GELU, rows=M, out8=hid8, c8_scale=8.0)
    h = hid8.view(torch.float8_e4m3fn).float()
    assert torch.isfinite(h).all() and float(h.abs().max()) == 448.0
    ref = torch.nn.functional.gelu(_deq(a8, asc, K) @ _deq(pw.hi, pw.lo, K).T + b.double()) * 8.0
    big = ref.abs() > 460
    assert bool(big.any()) and torch.equal(h[big], torch.sign(ref[big]).float() * 448.0)


def test_grouped_gathered_fp8_mlp_chain():
    """The expert MLP as the fp8 mode runs it: GEMM1 (gathered rows, grouped, GELU, e4m3 hidden with the static scale 8) then
    GEMM2 (uniform activation scale 1/8, gate-probability row scale), against fp64 on the same quantised tensors."""
    L, ops = pkg("_lib"), pkg("ops")
    D, F, G, S = 512, 1024, 5, 400
    sizes = [130, 0, 257, 1, 128]
    M = sum(sizes)
    goff = torch.tensor([0] + list(torch.tensor(sizes).cumsum(0)), dtype=torch.int32, device="cuda")
    src = _rand(S, D, seed=11, scale=2.0)
    g = torch.Generator(device="cpu").manual_seed(12)
    gather = torch.randint(0, S, (M,), generator=g, dtype=torch.int32).cuda()
    w1, b1 = _rand(G, F, D, seed=13, scale=D ** -0.5), _rand(G, F, seed=14, scale=0.1)
    w2, b2 = _rand(G, D, F, seed=15, scale=F ** -0.5), _rand(G, D, seed=16, scale=0.1)
    rs = _rand(M, seed=17).abs()
    a8, asc = ops.quantize_rows_fp8(src)
    p1, p2 = ops.PackedWeight(w1, fmt="f8"), ops.PackedWeight(w2, fmt="f8")
    hid8 = torch.zeros((M, F), dtype=torch.uint8, device="cuda")
    ops.gemm_fp8(a8, asc, p1, b1, act=L.ACT_GELU, gather=gather, goff=goff, rows=M, out8=hid8, c8_scale=8.0)
    out = torch.full((M + 2, D), 7.0, device="cuda")
    ops.gemm_fp8(hid8, None, p2, b2, a_scale_u=0.125, rowscale=rs, goff=goff, rows=M, out=out)
    x = _deq(a8, asc, D)[gather.long()]
    w1q, w2q = _deq(p1.hi, p1.lo, D).reshape(G, F, D), _deq(p2.hi, p2.lo, F).reshape(G, D, F)
    ref = torch.empty(M, D, dtype=torch.float64, device="cuda")
    o = 0
    for e, n in enumerate(sizes):
        h = torch.nn.functional.gelu(x[o:o + n] @ w1q[e].T + b1[e].double())
        h8 = (h.float() * 8.0).to(torch.float8_e4m3fn)
        got8 = hid8[o:o + n].view(torch.float8_e4m3fn).float()
        assert float((got8 != h8.float()).float().mean()) < 2e-3 if n else True  # only rounding-boundary ties may differ
        hq = hid8[o:o + n].view(torch.float8_e4m3fn).double() / 8.0
        ref[o:o + n] = (hq @ w2q[e].T + b2[e].double()) * rs[o:o + n, None].double()
        o += n
    assert rel_inf(out[:M].cpu(), ref.float().cpu()) < 2e-5
    assert torch.all(out[M:] == 7.0)


@pytest.mark.parametrize("Ln,T", [(2, 64), (4, 196)])
def test_configs4_big_16_experts_fp8_mode_error_and_flips(Ln, T):
    """Big widths (D 1024, F 2048, head_dim 256), E = 16, top-2, fp8 expert GEMMs, B = 8 -- at a reduced shape (L = 2, T = 64) and
    at BASELINE configs[4]'s real per-GPU shape (L = 4: 8 decoder layers, B = 64 / 8 GPUs = 8, T = 196): the fp8 mode against the
    oracle -- error and routing decisions that differ are printed and held to a budget; the fp16 mode on the same inputs is
    printed beside it so the cost of fp8 is visible."""
    T_, synth = pkg("transformer"), pkg("synth")
    B, E = 8, 16
    res = {}
    host = None
    for precision in (5, 2):
        m = T_.MotionTransformer(263, num_frames=196, latent_dim=512, ff_size=1024, num_layers=Ln, num_heads=4,
                                 text_latent_dim=256, moe_num_experts=E, model_size="big", precision=precision)
        if host is None:
            sd = synth.synth_state_dict(m._layout, 4)
            eph = synth.synth_ephemerals(1024, 512, Ln, 7)
            proj = synth.synth_projections(256, Ln, 7)
            host = (sd, eph, proj)
        sd, eph, proj = host
        m.load_state_dict(sd, strict=True)
        m.set_ephemerals(eph), m.set_projections(proj)
        m = m.cuda().eval()
        x, _, length, xf_proj, xf_out = synth.synth_inputs(B, T, 263, 28, 512, 4, min_len=24)
        t = torch.full((B,), 700, dtype=torch.int64)
        with RouteDump(2 * Ln, B, T) as rd:
            y = m(x.cuda(), t.cuda(), length.cuda(), xf_proj=xf_proj.cuda(), xf_out=xf_out.cuda()).cpu()
        if "ref" not in res:
            torch.set_num_threads(min(16, os.cpu_count() or 1))
            trace = {}
            with torch.no_grad():
                res["ref"] = R.denoiser_forward(sd, dict(latent_dim=1024, num_heads=4, num_layers=Ln, moe_num_experts=E), x, t,
                                                length, xf_proj, xf_out, {n: (w, b) for n, w, b in eph}, dict(proj), None, trace)
            res["trace"] = trace
        ref, trace = res["ref"], res["trace"]
        flips = decisions = 0
        forced = torch.zeros((2 * Ln, 2 * 2 * B * T), dtype=torch.int32)
        for li, name in enumerate(_layer_names(Ln)):
            S = T // 2 if li < Ln else T
            want = torch.stack([trace[f"{name}.ffn.branches.{br}.top2_idx"] for br in range(2)])
            forced[li, :want.numel()] = want.reshape(-1).to(torch.int32)
            flips += count_flips(rd.layer(li, B * S), want)[0]
            decisions += 2 * B * S
        # the same forward with the oracle's routing imposed: what is left is the arithmetic error of the mode
        yf = m(x.cuda(), t.cuda(), length.cuda(), xf_proj=xf_proj.cuda(), xf_out=xf_out.cuda(), forced_routing=forced).cpu()
        frame = (y - ref).abs().amax(-1) / ref.abs().max()
        frame_f = (yf - ref).abs().amax(-1) / ref.abs().max()
        res[precision] = (rel_inf(y, ref), float(frame.median()), flips, decisions, rel_inf(yf, ref), float(frame_f.median()))
        print(f"big / E=16 / L={Ln} T={T} / precision {precision}: rel err {res[precision][0]:.2e}, median frame err "
              f"{res[precision][1]:.2e}, routing decisions that differ {flips}/{decisions}; with the oracle's routing imposed: "
              f"rel err {res[precision][4]:.2e}, median frame err {res[precision][5]:.2e}")
        del m
        torch.cuda.empty_cache()
    assert torch.isfinite(torch.tensor(res[5][:2])).all()
    # Measured on MI355X.  L = 2, T = 64: fp8 4.4e-2 median frame error, 9 % of the routing decisions differ; fp16 beside it
    # 9.4e-4 / 0.1 %.  At the real depth (L = 4, T = 196) the e4m3 error compounds through the router: 29 % of the decisions
    # differ and the free-routing output is a different sample (median 3.6e-1); fp16: 2.1e-3 / 1.4 %.  The free-routing fp8 figures
    # at the real shape are therefore reported, and what is gated there is the arithmetic error under the oracle's routing
    # (fp8 8.8e-2 median / 2.1e-1 max; fp16 2.0e-3 / 5.1e-3).
    if (Ln, T) == (2, 64):
        assert res[5][1] < 8e-2 and res[5][2] <= 0.15 * res[5][3]
    else:
        assert res[5][5] < FP8_FORCED_MEDIAN_BUDGET and res[5][2] <= 0.40 * res[5][3]
    assert res[2][1] < 8e-3 and res[2][2] <= 0.03 * res[2][3]


@pytest.mark.parametrize("precision", [2, 5])
def test_sixteen_expert_router_with_compile_time_expert_count_is_bit_identical(precision):
    """E = 16, D = 1024 (BASELINE configs[4]): the router instantiated for the expert count and hn format (the default since round
    4, knob 26) against the run-time one (knob 27) through a whole forward at a ragged batch: same arithmetic in the same order, so
    bit-equal outputs (the fp8 rows are written after the logit loop and keep the run-time kernel: equal trivially)."""
    T_, synth = pkg("transformer"), pkg("synth")
    B, T, Ln, E = 3, 38, 1, 16
    m = T_.MotionTransformer(263, num_frames=196, latent_dim=512, ff_size=1024, num_layers=Ln, num_heads=4, text_latent_dim=256,
                             moe_num_experts=E, model_size="big", precision=precision)
    m.load_state_dict(synth.synth_state_dict(m._layout, 4), strict=True)
    m.set_ephemerals(synth.synth_ephemerals(1024, 512, Ln, 7)), m.set_projections(synth.synth_projections(256, Ln, 7))
    m = m.cuda().eval()
    x, _, length, xf_proj, xf_out = synth.synth_inputs(B, T, 263, 28, 512, 0, min_len=8)
    t = torch.full((B,), 500, dtype=torch.int64)
    lib = pkg("_lib").lib()
    outs = {}
    for knob in (27, 26, 0):
        lib.mdm_set_gemm_variant(knob)
        try:
            outs[knob] = m(x.cuda(), t.cuda(), length.cuda(), xf_proj=xf_proj.cuda(), xf_out=xf_out.cuda()).cpu()
        finally:
            lib.mdm_set_gemm_variant(0)
    assert torch.isfinite(outs[0]).all()
    assert torch.equal(outs[26], outs[27]), float((outs[26] - outs[27]).abs().max())
    assert torch.equal(outs[0], outs[27])
