"""GPU: each decoder block through mdm_block_forward against the oracle's restatement of the same block, in both
precision modes, at the real head_dim (128) with ragged lengths and S not a multiple of the tile sizes."""
import ctypes as C
import os
import sys

import pytest
import torch
import torch.nn.functional as F

from conftest import ROOT, build_module, load_golden, rel_inf, pkg

sys.path.insert(0, os.path.join(ROOT, "oracle"))
import denoiser_ref as R  # noqa: E402

pytestmark = pytest.mark.gpu
TOL = {3: 1e-3, 1: 3e-2, 2: 6e-3, 4: 1e-3}


def _setup(B, S, N, precision):
    g, meta = load_golden("fwd_small_dims")
    m, (sd, eph, proj, mcfg) = build_module(meta, precision=precision)
    synth = pkg("synth")
    D, Dt, H, E = 512, 256, 4, 8
    h = synth.uniform_pm1((B, S, D), "blk.h", S) * 1.5
    emb = synth.uniform_pm1((B, D), "blk.emb", S)
    xf = synth.uniform_pm1((B, N, Dt), "blk.xf", N) * 1.7
    length = torch.tensor([S, max(1, S - 13)][:B] + [S] * max(0, B - 2))
    pre = "decoder_blocks_low.0.module"
    sc = []
    for slot, sp in (("local_style", pre + ".dual_self_attn.local_attn.style_block"),
                     ("global_style", pre + ".dual_self_attn.global_attn.style_block"),
                     ("cross_style", pre + ".cross_attn.base_ca.proj_out"), ("ffn_style", pre + ".ffn.proj_out")):
        w, b = eph["low.0." + slot]
        e = F.linear(emb, w, b)
        sc.append(F.linear(F.silu(e), sd[sp + ".emb_layers.1.weight"], sd[sp + ".emb_layers.1.bias"]))
    sc = torch.stack(sc)  # (4, B, 2D)
    return m, sd, eph, proj, h, emb, xf, length, sc, pre, (D, H, E)


def _run_block(m, block, h, sc, length, xf, forced=None, ntok=None):
    L = pkg("_lib")
    pm = m.pack()
    B, S, D = h.shape
    tcache = m.prepare_text(xf.cuda(), ntok=ntok)
    ws = m._workspace(B, S, xf.shape[1])
    hd, scd, ld = h.cuda().contiguous(), sc.cuda().contiguous(), length.to(torch.int32).cuda()
    out = torch.empty_like(hd)
    fr = forced.to(torch.int32).cuda().contiguous() if forced is not None else None
    L.check(L.lib().mdm_block_forward(C.byref(pm.model), C.c_int32(0), C.c_int32(block), C.byref(tcache["tc"]),
                                      C.c_void_p(hd.data_ptr()), C.c_void_p(scd.data_ptr()), C.c_void_p(ld.data_ptr()),
                                      C.c_int32(B), C.c_int32(S), C.c_void_p(out.data_ptr()), C.c_void_p(ws.data_ptr()),
                                      C.c_int64(ws.numel()), C.c_void_p(L.ptr(fr)), C.c_int32(m.precision),
                                      C.c_void_p(L.stream_ptr())))
    return out.cpu()


@pytest.mark.parametrize("precision", [3, 1, 2, 4])
@pytest.mark.parametrize("S,N", [(40, 6), (98, 28), (196, 85)])
def test_blocks_match_oracle(S, N, precision):
    B = 2
    m, sd, eph, proj, h, emb, xf, length, sc, pre, (D, H, E) = _setup(B, S, N, precision)
    L = pkg("_lib")
    mask = R.src_mask(S, length)
    with torch.no_grad():
        ref_dual = R.dual_self_attention(h, emb, mask, sd, pre + ".dual_self_attn", H, eph, proj, "low.0")
        ref_cross = R.gated_cross_attention(h, xf, emb, sd, pre + ".cross_attn", H, eph["low.0.cross_style"])
        trace = {}
        ref_moe = R.moe_ffn(h, emb, sd, pre + ".ffn", E, eph["low.0.ffn_style"], None, trace)
        ref_sd = R.softmax_cross_ffn(h, xf, sd, pre + ".sd_cross_attn", H)
    forced = torch.stack([trace[f"{pre}.ffn.branches.{b}.top2_idx"] for b in range(2)])  # (2, M, 2)
    errs = {
        "dual": rel_inf(_run_block(m, L.BLOCK_DUAL, h, sc, length, xf), ref_dual),
        "cross": rel_inf(_run_block(m, L.BLOCK_CROSS, h, sc, length, xf), ref_cross),
        "moe": rel_inf(_run_block(m, L.BLOCK_MOE, h, sc, length, xf, forced if precision in (1, 2) else None), ref_moe),
        "sdcross": rel_inf(_run_block(m, L.BLOCK_SDCROSS, h, sc, length, xf), ref_sd),
    }
    print(f"S={S} N={N} precision={precision}:", {k: f"{v:.2e}" for k, v in errs.items()})
    bad = {k: v for k, v in errs.items() if not v < TOL[precision]}
    assert not bad, bad


def test_free_routing_flip_budget_bf16():
    """Throughput mode with FREE routing: gate logits are fp32 (router kernel) computed from an fp32 LayerNorm, so the
    top-2 choice only differs from the oracle's where p2 - p3 is within fp32 noise: none expected at this size."""
    B, S, N = 2, 98, 6
    m, sd, eph, proj, h, emb, xf, length, sc, pre, (D, H, E) = _setup(B, S, N, 1)
    L = pkg("_lib")
    with torch.no_grad():
        ref_moe = R.moe_ffn(h, emb, sd, pre + ".ffn", E, eph["low.0.ffn_style"], None, None)
    out = _run_block(m, L.BLOCK_MOE, h, sc, length, xf)
    tok_err = (out - ref_moe).abs().amax(-1) / ref_moe.abs().amax()
    flipped = int((tok_err > 0.1).sum())
    print("tokens with O(1) change (routing flips):", flipped, "of", B * S)
    assert flipped <= 1


@pytest.mark.parametrize("precision", [2, 1])
@pytest.mark.parametrize("B,S", [(2, 98), (3, 37), (1, 5)])
def test_router_with_compile_time_expert_count_is_bit_identical(B, S, precision):
    """At D = 512, E = 8 the 16-bit modes run the router instantiated for that expert count and format (straight-line
    chunk loop, clamped instead of predicated hn stores); knob 27 selects the run-time-E kernel, knob 26 the constant one
    wherever it exists.  Same arithmetic in the same order: the MoE block's outputs must be bit-equal, also where B * S is
    not a multiple of the 16 tokens a workgroup iteration takes."""
    m, sd, eph, proj, h, emb, xf, length, sc, pre, (D, H, E) = _setup(B, S, 6, precision)
    L = pkg("_lib")
    out = _run_block(m, L.BLOCK_MOE, h, sc, length, xf)
    outs = {}
    for knob in (27, 26):
        L.lib().mdm_set_gemm_variant(knob)
        try:
            outs[knob] = _run_block(m, L.BLOCK_MOE, h, sc, length, xf)
        finally:
            L.lib().mdm_set_gemm_variant(0)
    ref = outs[27]
    assert torch.equal(outs[26], ref), float((outs[26] - ref).abs().max())
    assert torch.equal(out, ref), float((out - ref).abs().max())
    assert torch.isfinite(out).all()
    assert torch.equal(out, ref), float((out - ref).abs().max())


@pytest.mark.parametrize("S,N,B", [(196, 28, 3), (98, 28, 2), (50, 9, 1), (12, 32, 2), (196, 85, 2), (98, 33, 2), (40, 64, 1), (77, 43, 2),
                                    (30, 128, 1)])
def test_sd_fold_matches_unfolded_chain(S, N, B):
    """Throughput mode: the text cross-attention with folded projections (csrc/sdfold.hip, one launch) against the
    oracle and against the unfolded chain (query GEMM, attention core, output GEMM, LayerNorm; kernel knob 22).  N > 32
    text tokens take several passes of whole heads (2 at N = 33..64 with 4 heads, 4 up to 128; the reference pads captions
    to 8 + 77 = 85 tokens, text_encoder.py:19,26); by default the module folds up to two passes and keeps the chain beyond,
    where it measures faster -- knob 24 forces the fold."""
    m, sd, eph, proj, h, emb, xf, length, sc, pre, (D, H, E) = _setup(B, S, N, 1)
    L = pkg("_lib")
    with torch.no_grad():
        ref = R.softmax_cross_ffn(h, xf, sd, pre + ".sd_cross_attn", H)
    L.lib().mdm_set_gemm_variant(24)  # fold at any supported N (by default the chain is taken beyond two passes, where it is faster)
    try:
        folded = _run_block(m, L.BLOCK_SDCROSS, h, sc, length, xf)
    finally:
        L.lib().mdm_set_gemm_variant(0)
    L.lib().mdm_set_gemm_variant(22)
    try:
        chain = _run_block(m, L.BLOCK_SDCROSS, h, sc, length, xf)
    finally:
        L.lib().mdm_set_gemm_variant(0)
    e_f, e_c, d = rel_inf(folded, ref), rel_inf(chain, ref), rel_inf(folded, chain)
    print(f"S={S} N={N}: folded {e_f:.2e}  chain {e_c:.2e}  folded-vs-chain {d:.2e}")
    assert e_f < TOL[1] and e_c < TOL[1]
    assert not torch.equal(folded, chain)  # the knob really selects two different code paths


@pytest.mark.parametrize("precision,knob", [(3, 0), (3, 56), (4, 0), (2, 0), (2, 22), (1, 24), (2, 24)])
@pytest.mark.parametrize("S,N,counts", [(98, 28, [28, 9, 17]), (40, 64, [1, 64, 33]), (196, 85, [85, 10, 47])])
def test_per_sample_token_counts_equal_each_sample_run_with_its_own_tokens(S, N, counts, precision, knob):
    """MdmTextCache.ntok: samples whose captions have different token counts travel in one batch, padded to N rows; both
    text cross-attentions must give every sample what it gets when run alone with exactly its own tokens (what the reference
    computes: it has no text mask, so it can only run such samples in separate forwards), whatever the padding rows hold.
    Paths: fp32-grade one-launch cores (csrc/xattn3.hip) and their chain (knob 56: row softmax), folded kernel (knob 24 forces it), attention-core kernel (knob 22), and against
    the oracle on the sample's own tokens."""
    B = len(counts)
    m, sd, eph, proj, h, emb, xf, length, sc, pre, (D, H, E) = _setup(B, S, N, precision)
    L = pkg("_lib")
    synth = pkg("synth")
    length = torch.tensor([S, max(1, S - 13), S][:B])
    pad = synth.uniform_pm1(tuple(xf.shape), "blk.pad", N) * 9.0   # the padding rows hold junk, not zeros
    xf_pad = xf.clone()
    for b, n in enumerate(counts):
        xf_pad[b, n:] = pad[b, n:]
    tol = {3: 2e-5, 4: 2e-5, 2: 2e-3, 1: 1.5e-2}[precision]
    L.lib().mdm_set_gemm_variant(knob)
    try:
        for block, ref_fn in ((L.BLOCK_CROSS, "cross"), (L.BLOCK_SDCROSS, "sd")):
            got = _run_block(m, block, h, sc, length, xf_pad, ntok=counts)
            assert torch.isfinite(got).all()
            for b, n in enumerate(counts):
                sl = slice(b, b + 1)
                alone = _run_block(m, block, h[sl], sc[:, sl], length[sl], xf[sl, :n].contiguous())
                with torch.no_grad():
                    if ref_fn == "cross":
                        ref = R.gated_cross_attention(h[sl], xf[sl, :n], emb[sl], sd, pre + ".cross_attn", H, eph["low.0.cross_style"])
                    else:
                        ref = R.softmax_cross_ffn(h[sl], xf[sl, :n], sd, pre + ".sd_cross_attn", H)
                e_alone, e_ref = rel_inf(got[sl], alone), rel_inf(got[sl], ref)
                print(f"{ref_fn} S={S} N={N} n={n} precision {precision} knob {knob}: vs alone {e_alone:.2e}, vs oracle {e_ref:.2e}")
                assert e_alone < tol and e_ref < TOL[precision]
        # an unmasked run over the junk rows must differ: the counts are really applied
        junk = _run_block(m, L.BLOCK_SDCROSS, h, sc, length, xf_pad)
        assert rel_inf(junk, got) > 10 * tol or all(n == N for n in counts)
    finally:
        L.lib().mdm_set_gemm_variant(0)
    with pytest.raises(ValueError):
        m.prepare_text(xf_pad.cuda(), ntok=[0] * B)


@pytest.mark.parametrize("precision", [2, 1])
@pytest.mark.parametrize("B,S", [(2, 98), (3, 37), (1, 5), (5, 196)])
def test_performer_tail_fused_into_the_projection_pair(B, S, precision):
    """16-bit modes, D = 512: the Performer's proj_out pair, post_norm, stylization, out_layers.2, the residual and the pre_norm of
    the global branch run in ONE launch (csrc/mlp_stream.hip pair_tail); knob 35 runs the tail as its own launches (style_gemm +
    ln_chain).  Same arithmetic per row: the block's outputs agree to rounding (a last-bit difference in front of the 16-bit image
    can flip one 16-bit rounding: 2^-11 of one of 512 terms in fp16, 2^-8 in bf16), also where the last tile is ragged (B * S not
    a multiple of 16 / 32 / 64), and both agree with the oracle."""
    m, sd, eph, proj, h, emb, xf, length, sc, pre, (D, H, E) = _setup(B, S, 6, precision)
    L = pkg("_lib")
    fused = _run_block(m, L.BLOCK_DUAL, h, sc, length, xf)
    L.lib().mdm_set_gemm_variant(35)
    try:
        split = _run_block(m, L.BLOCK_DUAL, h, sc, length, xf)
    finally:
        L.lib().mdm_set_gemm_variant(0)
    with torch.no_grad():
        ref = R.dual_self_attention(h, emb, R.src_mask(S, length), sd, pre + ".dual_self_attn", H, eph, proj, "low.0")
    d, e_f, e_s = rel_inf(fused, split), rel_inf(fused, ref), rel_inf(split, ref)
    print(f"B={B} S={S} precision {precision}: fused vs split {d:.2e}; vs oracle {e_f:.2e} / {e_s:.2e}")
    assert torch.isfinite(fused).all()
    assert d < (2e-4 if precision == 2 else 2e-3)
    assert e_f < TOL[precision] and e_s < TOL[precision]


@pytest.mark.parametrize("precision", [2, 1])
@pytest.mark.parametrize("B,S", [(2, 98), (3, 37), (1, 5), (2, 196), (4, 208)])
def test_qkv_projection_inside_the_attention_core(B, S, precision):
    """16-bit modes, head_dim 128: the Performer's query | key | value projection runs inside the attention core's launch
    (csrc/perf_attn.hip phase 0); knob 50 runs it as its own GEMM launch.  Same products, same 16-bit rounding of q | k | v:
    the block outputs agree to accumulation-order rounding, and both agree with the oracle; ragged lengths, S not a multiple
    of 16, and the largest S the fused form takes (208)."""
    m, sd, eph, proj, h, emb, xf, length, sc, pre, (D, H, E) = _setup(B, S, 6, precision)
    L = pkg("_lib")
    fused = _run_block(m, L.BLOCK_DUAL, h, sc, length, xf)
    L.lib().mdm_set_gemm_variant(50)
    try:
        split = _run_block(m, L.BLOCK_DUAL, h, sc, length, xf)
    finally:
        L.lib().mdm_set_gemm_variant(0)
    with torch.no_grad():
        ref = R.dual_self_attention(h, emb, R.src_mask(S, length), sd, pre + ".dual_self_attn", H, eph, proj, "low.0")
    d, e_f, e_s = rel_inf(fused, split), rel_inf(fused, ref), rel_inf(split, ref)
    print(f"B={B} S={S} precision {precision}: qkv inside vs own launch {d:.2e}; vs oracle {e_f:.2e} / {e_s:.2e}")
    assert torch.isfinite(fused).all()
    assert d < (1e-3 if precision == 2 else 8e-3)
    assert e_f < TOL[precision] and e_s < TOL[precision]


@pytest.mark.parametrize("precision", [2, 1])
@pytest.mark.parametrize("B,S,N", [(2, 98, 28), (3, 37, 6), (1, 5, 9), (2, 196, 28), (2, 208, 85)])
def test_query_projection_inside_the_linear_cross_attention(B, S, N, precision):
    """16-bit modes, head_dim 128: the query projection of GatedCrossAttention runs inside the attention launch
    (csrc/xattn.hip lin_xattn_q); knob 51 runs it as its own GEMM launch.  Same products, same 16-bit rounding of q."""
    m, sd, eph, proj, h, emb, xf, length, sc, pre, (D, H, E) = _setup(B, S, N, precision)
    L = pkg("_lib")
    fused = _run_block(m, L.BLOCK_CROSS, h, sc, length, xf)
    L.lib().mdm_set_gemm_variant(51)
    try:
        split = _run_block(m, L.BLOCK_CROSS, h, sc, length, xf)
    finally:
        L.lib().mdm_set_gemm_variant(0)
    with torch.no_grad():
        ref = R.gated_cross_attention(h, xf, emb, sd, pre + ".cross_attn", H, eph["low.0.cross_style"])
    d, e_f, e_s = rel_inf(fused, split), rel_inf(fused, ref), rel_inf(split, ref)
    print(f"B={B} S={S} N={N} precision {precision}: query inside vs own launch {d:.2e}; vs oracle {e_f:.2e} / {e_s:.2e}")
    assert torch.isfinite(fused).all()
    assert d < (1e-3 if precision == 2 else 8e-3)
    assert e_f < TOL[precision] and e_s < TOL[precision]


@pytest.mark.parametrize("precision", [3, 1, 2, 4])
def test_named_block_entry_points(precision):
    """The per-block C entry points named in SURVEY.md §8(b): the aliases must reproduce mdm_block_forward bit for bit, and
    mdm_performer_attn_forward (one PerformerSelfAttention) is checked against the oracle for both attention slots."""
    B, S, N = 2, 98, 28
    m, sd, eph, proj, h, emb, xf, length, sc, pre, (D, H, E) = _setup(B, S, N, precision)
    L = pkg("_lib")
    lib = L.lib()
    pm = m.pack()
    tcache = m.prepare_text(xf.cuda())
    ws = m._workspace(B, S, N)
    hd, scd, ld = h.cuda().contiguous(), sc.cuda().contiguous(), length.to(torch.int32).cuda()
    common = (C.c_void_p(hd.data_ptr()), C.c_void_p(scd.data_ptr()), C.c_void_p(ld.data_ptr()), C.c_int32(B), C.c_int32(S))
    tail = (C.c_void_p(ws.data_ptr()), C.c_int64(ws.numel()))

    def call(fn, *pre_args, extra=()):
        out = torch.empty_like(hd)
        L.check(fn(C.byref(pm.model), C.c_int32(0), *pre_args, *common, C.c_void_p(out.data_ptr()), *tail, *extra,
                   C.c_int32(precision), C.c_void_p(L.stream_ptr())))
        return out.cpu()

    tc = C.byref(tcache["tc"])
    null = C.c_void_p(0)
    m.reset_all_moe_counters(m)
    a_moe = call(lib.mdm_moe_ffn_forward, extra=(null,))
    assert torch.equal(call(lib.mdm_dual_self_attn_forward), _run_block(m, L.BLOCK_DUAL, h, sc, length, xf))
    assert torch.equal(call(lib.mdm_linear_xattn_forward, tc), _run_block(m, L.BLOCK_CROSS, h, sc, length, xf))
    assert torch.equal(call(lib.mdm_softmax_xattn_ffn_forward, tc), _run_block(m, L.BLOCK_SDCROSS, h, sc, length, xf))
    assert torch.equal(a_moe, _run_block(m, L.BLOCK_MOE, h, sc, length, xf))
    mask = R.src_mask(S, length)
    for which, slot in ((0, "local"), (1, "global")):
        out = torch.empty_like(hd)
        scw = sc[which].cuda().contiguous()
        L.check(lib.mdm_performer_attn_forward(C.byref(pm.model), C.c_int32(0), C.c_int32(which), C.c_void_p(hd.data_ptr()),
                                               C.c_void_p(scw.data_ptr()), C.c_void_p(ld.data_ptr()), C.c_int32(B),
                                               C.c_int32(S), C.c_void_p(out.data_ptr()), *tail, C.c_int32(precision),
                                               C.c_void_p(L.stream_ptr())))
        with torch.no_grad():
            ref = R.performer_self_attention(h, emb, mask, sd, f"{pre}.dual_self_attn.{slot}_attn", H,
                                             eph[f"low.0.{slot}_style"], proj[f"low.0.{slot}"])
        err = rel_inf(out.cpu(), ref)
        print(f"performer {slot} precision {precision}: {err:.2e}")
        assert err < TOL[precision], (slot, err)


@pytest.mark.parametrize("precision", [3, 4])
@pytest.mark.parametrize("B,S", [(2, 98), (3, 196), (2, 40), (1, 6), (2, 224)])
def test_fp32_grade_attention_core_in_one_launch(B, S, precision):
    """csrc/perf_attn3.hip (+ the ACT_HEADNORM epilogue of the q | k | v projection, csrc/gemm3.hip): the Performer core of
    the fp32-grade modes as ONE launch on bf16x3 products against (a) the five-launch chain it replaces (knob 52: head_norm,
    feature GEMM, KV-state GEMM, numerator GEMM, den_ln) on identical inputs and (b) the oracle's block; ragged lengths (a
    masked tail, a length-1 sample), frame counts that are not multiples of the 16-frame tile or the 32-frame chunk, one
    chunk (S <= 32) and the largest supported S."""
    m, sd, eph, proj, h, emb, xf, length, sc, pre, (D, H, E) = _setup(B, S, 6, precision)
    L = pkg("_lib")
    length = length.clone()
    if B >= 2:
        length[1] = max(1, S - 13)
    if B >= 3:
        length[2] = 1
    mask = R.src_mask(S, length)
    with torch.no_grad():
        ref = R.dual_self_attention(h, emb, mask, sd, pre + ".dual_self_attn", H, eph, proj, "low.0")
    out = _run_block(m, L.BLOCK_DUAL, h, sc, length, xf)
    L.lib().mdm_set_gemm_variant(52)
    try:
        chain = _run_block(m, L.BLOCK_DUAL, h, sc, length, xf)
    finally:
        L.lib().mdm_set_gemm_variant(0)
    e_ref, e_chain = rel_inf(out, ref), rel_inf(out, chain)
    print(f"B={B} S={S} precision={precision}: fused vs oracle {e_ref:.2e}, chain vs oracle {rel_inf(chain, ref):.2e}, fused vs chain {e_chain:.2e}")
    assert torch.isfinite(out).all()
    assert e_ref < 1e-3 and e_chain < 2e-4


@pytest.mark.parametrize("precision", [3, 4])
@pytest.mark.parametrize("B,S,N", [(2, 98, 28), (3, 196, 85), (2, 40, 6), (1, 6, 128), (2, 196, 33), (2, 50, 64)])
def test_fp32_grade_text_cross_attention_cores_in_one_launch(B, S, N, precision):
    """csrc/xattn3.hip behind the plane / head-softmax epilogues of the query projections (csrc/gemm3.hip): the linear and the
    softmax text cross-attention of the fp32-grade modes against (a) the chains they replace (knob 56) and (b) the oracle;
    1, 2, 3 and 4 key steps of 32 tokens, frame counts that are not multiples of 16."""
    m, sd, eph, proj, h, emb, xf, length, sc, pre, (D, H, E) = _setup(B, S, N, precision)
    L = pkg("_lib")
    with torch.no_grad():
        refs = {L.BLOCK_CROSS: R.gated_cross_attention(h, xf, emb, sd, pre + ".cross_attn", H, eph["low.0.cross_style"]),
                L.BLOCK_SDCROSS: R.softmax_cross_ffn(h, xf, sd, pre + ".sd_cross_attn", H)}
    for block, ref in refs.items():
        out = _run_block(m, block, h, sc, length, xf)
        L.lib().mdm_set_gemm_variant(56)
        try:
            chain = _run_block(m, block, h, sc, length, xf)
        finally:
            L.lib().mdm_set_gemm_variant(0)
        e_ref, e_chain = rel_inf(out, ref), rel_inf(out, chain)
        print(f"block {block} B={B} S={S} N={N} precision={precision}: fused vs oracle {e_ref:.2e}, chain vs oracle {rel_inf(chain, ref):.2e}, fused vs chain {e_chain:.2e}")
        assert torch.isfinite(out).all()
        assert e_ref < 1e-3 and e_chain < 2e-4  # (the two paths may agree bit for bit: same split products, same k order)


@pytest.mark.parametrize("precision", [3, 4])
@pytest.mark.parametrize("B,S", [(2, 98), (3, 196), (1, 5), (5, 37), (2, 130)])
def test_fp32_grade_stylization_in_one_launch(B, S, precision):
    """csrc/style_gemm.hip style_gemm3: stylization input (post-norm / MoE combine / plain), its D x D Linear on bf16x3 products
    with the (hi, lo) pair stream, and the residual in ONE launch, against the two launches it replaces (knob 60: style_in + GEMM) and
    the oracle, through the blocks that end in a StylizationBlock; ragged last tiles."""
    m, sd, eph, proj, h, emb, xf, length, sc, pre, (D, H, E) = _setup(B, S, 6, precision)
    L = pkg("_lib")
    mask = R.src_mask(S, length)
    with torch.no_grad():
        trace = {}
        refs = {L.BLOCK_DUAL: R.dual_self_attention(h, emb, mask, sd, pre + ".dual_self_attn", H, eph, proj, "low.0"),
                L.BLOCK_CROSS: R.gated_cross_attention(h, xf, emb, sd, pre + ".cross_attn", H, eph["low.0.cross_style"]),
                L.BLOCK_MOE: R.moe_ffn(h, emb, sd, pre + ".ffn", E, eph["low.0.ffn_style"], None, trace)}
    for block, ref in refs.items():
        out = _run_block(m, block, h, sc, length, xf)
        alt = {}
        for knob in (60, 61):  # 60: style_in + GEMM as two launches; 61: fused, but the LayerNorms / block tail behind it as launches
            L.lib().mdm_set_gemm_variant(knob)
            try:
                alt[knob] = _run_block(m, block, h, sc, length, xf)
            finally:
                L.lib().mdm_set_gemm_variant(0)
        e_ref, e_two, e_ln = rel_inf(out, ref), rel_inf(out, alt[60]), rel_inf(out, alt[61])
        print(f"block {block} B={B} S={S} precision={precision}: fused vs oracle {e_ref:.2e}, two launches vs oracle {rel_inf(alt[60], ref):.2e}, "
              f"fused vs two {e_two:.2e}, vs fused-without-tails {e_ln:.2e}")
        assert torch.isfinite(out).all()
        assert e_ref < 1e-3 and e_two < 2e-4 and e_ln < 2e-4


@pytest.mark.parametrize("precision", [3, 4])
@pytest.mark.parametrize("B,S,N", [(2, 98, 28), (3, 196, 6), (1, 5, 85)])
def test_pre_split_rows_between_fp32_grade_gemms_change_nothing(B, S, N, precision):
    """MDM_OP_X2_ROW: in the fp32-grade modes a tensor whose only consumer is a bf16x3 GEMM is written pre-split (bf16 hi | lo per
    block of 32 columns) by its producer -- LayerNorm kernels, attention cores, the router's LN rows, GEMM epilogues -- so that the
    GEMM does not re-split every fp32 fragment in its K loop.  The split is the same function of the same fp32 value wherever it
    happens, so every block's output must be bit-identical with the plumbing switched off (knob 62)."""
    m, sd, eph, proj, h, emb, xf, length, sc, pre, (D, H, E) = _setup(B, S, N, precision)
    L = pkg("_lib")
    for block in (L.BLOCK_DUAL, L.BLOCK_CROSS, L.BLOCK_MOE, L.BLOCK_SDCROSS, L.BLOCK_LAYER):
        out = _run_block(m, block, h, sc, length, xf)
        L.lib().mdm_set_gemm_variant(62)
        try:
            plain = _run_block(m, block, h, sc, length, xf)
        finally:
            L.lib().mdm_set_gemm_variant(0)
        assert torch.isfinite(out).all()
        assert torch.equal(out, plain), (block, float((out - plain).abs().max()))


@pytest.mark.parametrize("B,S", [(2, 98), (3, 196), (1, 5)])
def test_fp32_grade_expert_gemms_on_the_streamed_kernel_change_nothing(B, S):
    """csrc/gemm_stream3.hip inside the model (MdmPacked.ws of the expert matrices, fp32-grade mode): the MoE block and the whole
    decoder layer must be bit-identical with the expert GEMM pair on the 128 x 128 tile kernel (knob 69) -- same products, same
    order -- at group sizes from empty to several tiles (the router decides them)."""
    m, sd, eph, proj, h, emb, xf, length, sc, pre, (D, H, E) = _setup(B, S, 28, 3)
    L = pkg("_lib")
    assert any(k.endswith("w1") for k in m.pack().wstream1), "the fp32-grade model packs pair streams for its expert matrices"
    for block in (L.BLOCK_MOE, L.BLOCK_LAYER):
        out = _run_block(m, block, h, sc, length, xf)
        L.lib().mdm_set_gemm_variant(69)
        try:
            tile = _run_block(m, block, h, sc, length, xf)
        finally:
            L.lib().mdm_set_gemm_variant(0)
        assert torch.isfinite(out).all() and torch.equal(out, tile)
