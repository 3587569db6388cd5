"""GPU: the fused MFMA GEMM (csrc/gemm.hip) through the C ABI against a torch fp64 reference."""
import ctypes as C

import pytest
import torch

from conftest import pkg, rel_inf

pytestmark = pytest.mark.gpu


def _mods():
    return pkg("_lib"), pkg("ops")


def _rand(*shape, seed=0, dev="cuda"):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1).to(dev)


TOL = {1: 2e-2, 3: 2e-5}


@pytest.mark.parametrize("precision", [1, 3])
@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (256, 384, 512), (200, 263, 512), (70, 512, 263), (1, 96, 40),
                                   (392, 1536, 512)])
def test_linear_plain(M, N, K, precision):
    L, ops = _mods()
    x, w, b = _rand(M, K, seed=1), _rand(N, K, seed=2), _rand(N, seed=3)
    pw = ops.PackedWeight(w)
    y = ops.linear(x, pw, b, precision=precision)
    ref = (x.double() @ w.double().T + b.double()).float()
    assert rel_inf(y.cpu(), ref.cpu()) < TOL[precision]


def test_asymmetric_identity_layout():
    """A = I with asymmetric W catches row/col swaps in the MFMA C/D map."""
    L, ops = _mods()
    n = 128
    x = torch.eye(n, device="cuda")
    w = (torch.arange(n * n, device="cuda", dtype=torch.float32).reshape(n, n) % 251) - 125.0  # exact in bf16
    y = ops.linear(x, ops.PackedWeight(w), None, precision=1)
    assert torch.equal(y, w.T.contiguous())


@pytest.mark.parametrize("precision", [1, 3])
def test_epilogue_options(precision):
    L, ops = _mods()
    M, N, K, T = 150, 200, 96, 50
    x, w, b = _rand(M, K, seed=4), _rand(N, K, seed=5), _rand(N, seed=6)
    cs, rs = _rand(N, seed=7), _rand(M, seed=8)
    r1, r2 = _rand(T, N, seed=9), _rand(M, N, seed=10)
    pw = ops.PackedWeight(w)
    lin = x.double() @ w.double().T + b.double()
    for act, fn in [(L.ACT_NONE, lambda v: v), (L.ACT_GELU, lambda v: torch.nn.functional.gelu(v)),
                    (L.ACT_SILU, lambda v: torch.nn.functional.silu(v)),
                    (L.ACT_FEAT, lambda v: 0.1 * torch.exp(v.clamp(-15, 15)))]:
        y = ops.linear(x, pw, b, act=act, alpha=0.7, out_scale=0.3, colscale=cs, rowscale=rs, r1=r1, r1_scale=0.25,
                       r1_mod=T, r2=r2, precision=precision)
        ref = fn(0.7 * lin) * 0.3 * cs.double()[None] * rs.double()[:, None]
        ref = ref + 0.25 * r1.double()[torch.arange(M, device="cuda") % T] + r2.double()
        assert rel_inf(y.cpu(), ref.float().cpu()) < TOL[precision], act


@pytest.mark.parametrize("precision", [1, 3])
def test_batched_kstride_and_rowmajor_operands(precision):
    """The Performer contractions: KV^T[b,h] = 0.1 * V^T Kf (both operands K-strided) and num = 0.1 * Qf KV."""
    L, ops = _mods()
    B, H, S, dh, m = 3, 4, 52, 128, 128
    D = H * dh
    qkv = _rand(B * S, 3 * D, seed=11)
    phi = _rand(B * S, 2 * H * m, seed=12).abs()
    kvt = torch.empty(B, H, dh, m, device="cuda")
    d = ops.gemm_desc(precision)
    d.A = ops.f32_operand(qkv, 3 * D, L.OP_F32_KSTRIDE, offset=2 * D)
    d.A.bs1, d.A.bs2 = S * 3 * D, dh
    d.W = ops.f32_operand(phi, 2 * H * m, L.OP_F32_KSTRIDE, offset=H * m)
    d.W.bs1, d.W.bs2 = S * 2 * H * m, m
    d.M, d.N, d.K = dh, m, S
    d.batch, d.nb2 = B * H, H
    d.C, d.ldc, d.c_bs1, d.c_bs2 = kvt.data_ptr(), m, H * dh * m, dh * m
    d.out_scale = 0.1
    ops.run_gemm(d)
    v = qkv.view(B, S, 3, H, dh)[:, :, 2].double()  # (B,S,H,dh)
    kf = phi.view(B, S, 2, H, m)[:, :, 1].double()
    ref = 0.1 * torch.einsum("bshd,bshm->bhdm", v, kf)
    assert rel_inf(kvt.cpu(), ref.float().cpu()) < TOL[precision]
    # num[b,s,h,:] = 0.1 * qf[b,s,h,:] @ KV[b,h]  with W = KV^T rows (dh x m), row-major fp32
    num = torch.empty(B * S, D, device="cuda")
    d = ops.gemm_desc(precision)
    d.A = ops.f32_operand(phi, 2 * H * m)
    d.A.bs1, d.A.bs2 = S * 2 * H * m, m
    d.W = ops.f32_operand(kvt, m)
    d.W.bs1, d.W.bs2 = H * dh * m, dh * m
    d.M, d.N, d.K = S, dh, m
    d.batch, d.nb2 = B * H, H
    d.C, d.ldc, d.c_bs1, d.c_bs2 = num.data_ptr(), D, S * D, dh
    d.out_scale = 0.1
    ops.run_gemm(d)
    qf = phi.view(B, S, 2, H, m)[:, :, 0].double()
    ref2 = 0.1 * torch.einsum("bshm,bhdm->bshd", qf, kvt.double()).reshape(B * S, D)
    assert rel_inf(num.cpu(), ref2.float().cpu()) < TOL[precision]


@pytest.mark.parametrize("precision", [1, 3])
def test_grouped_gather_feature_rows(precision):
    """Expert-style grouped GEMM with gathered rows + grouped rows-per-token addressing with key masking."""
    L, ops = _mods()
    E, D, F_, Mtok = 5, 96, 160, 333
    x = _rand(Mtok, D, seed=13)
    w = _rand(E, F_, D, seed=14)
    b = _rand(E, F_, seed=15)
    g = torch.Generator().manual_seed(3)
    counts = torch.tensor([0, 130, 1, 257, 61])
    off = torch.zeros(E + 1, dtype=torch.int32)
    off[1:] = counts.cumsum(0)
    tot = int(off[-1])
    gather = torch.randint(0, Mtok, (tot,), generator=g, dtype=torch.int32)
    pw = ops.PackedWeight(w)
    out = torch.zeros(tot, F_, device="cuda")
    d = ops.gemm_desc(precision)
    d.A = ops.f32_operand(x, D)
    gd = gather.cuda()
    d.A.gather = gd.data_ptr()
    d.W = pw.operand()
    d.W.bs1 = F_ * pw.Kp
    od = off.cuda()
    d.goff, d.ngroups = od.data_ptr(), E
    d.M, d.N, d.K = tot, F_, D
    bd = b.contiguous()
    d.bias, d.bias_bs = bd.data_ptr(), F_
    d.C, d.ldc = out.data_ptr(), F_
    d.act = L.ACT_GELU
    ops.run_gemm(d)
    ref = torch.empty(tot, F_, dtype=torch.float64)
    for e in range(E):
        r = slice(int(off[e]), int(off[e + 1]))
        xe = x.cpu().double()[gather[r].long()]
        ref[r] = torch.nn.functional.gelu(xe @ w[e].cpu().double().T + b[e].cpu().double())
    assert rel_inf(out.cpu(), ref.float()) < TOL[precision]

    # feature map over (token, slot) rows: slots [0,H) are queries, [H,2H) keys, masked past length
    B, S, H, dh, m = 2, 10, 2, 32, 32
    qkv = _rand(B * S, 3 * H * dh, seed=16)
    P = _rand(dh, m, seed=17) * 0.3
    pw = ops.PackedWeight(P.T.contiguous())
    lens = torch.tensor([10, 6], dtype=torch.int32, device="cuda")
    phi = torch.empty(B * S, 2 * H, m, device="cuda")
    d = ops.gemm_desc(precision)
    d.A = ops.f32_operand(qkv, dh)
    d.A.rpg, d.A.gstride = 2 * H, 3 * H * dh
    d.W = pw.operand()
    d.M, d.N, d.K = B * S * 2 * H, m, dh
    d.C, d.ldc = phi.data_ptr(), m
    d.act = L.ACT_FEAT
    d.feat_len, d.feat_S, d.feat_rpt, d.feat_kslot = lens.data_ptr(), S, 2 * H, H
    ops.run_gemm(d)
    z = qkv.view(B, S, 3, H, dh)[:, :, :2].double() @ P.double()
    ref = 0.1 * torch.exp(z.clamp(-15, 15))
    mask = (torch.arange(S, device="cuda")[None] < lens[:, None]).double()
    ref[:, :, 1] *= mask[:, :, None, None]
    assert rel_inf(phi.cpu(), ref.reshape(B * S, 2 * H, m).float().cpu()) < TOL[precision]


def test_gemm_rejects_bad_args():
    L, ops = _mods()
    d = ops.gemm_desc(1)
    assert L.lib().mdm_gemm(C.byref(d), C.c_void_p(0)) == 0  # empty problem is a no-op
    d.M = d.N = d.K = 32
    assert L.lib().mdm_gemm(C.byref(d), C.c_void_p(0)) == 1  # null operands


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (12544, 512, 512), (200, 263, 512), (70, 512, 1024), (1, 96, 64),
                                   (392, 1536, 512)])
def test_bf16_activation_kernel(M, N, K):
    """gemm2.hip: bf16 A via LDS-DMA, swapped-operand epilogue; result must equal the fp32-A kernel fed the same
    bf16-rounded inputs (both accumulate bf16 products in fp32)."""
    L, ops = _mods()
    x, w, b = _rand(M, K, seed=21), _rand(N, K, seed=22), _rand(N, seed=23)
    xb = x.to(torch.bfloat16)
    pw = ops.PackedWeight(w)
    r1, r2, cs, rs = _rand(M, N, seed=24), _rand(M, N, seed=25), _rand(N, seed=26), _rand(M, seed=27)
    out16 = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    y = ops.linear(xb, pw, b, act=L.ACT_GELU, alpha=0.9, out_scale=0.5, colscale=cs, rowscale=rs, r1=r1, r1_scale=0.3,
                   r2=r2, precision=1, out16=out16)
    ref = torch.nn.functional.gelu(0.9 * (xb.double() @ w.to(torch.bfloat16).double().T + b.double())) * 0.5
    ref = ref * cs.double()[None] * rs.double()[:, None] + 0.3 * r1.double() + r2.double()
    assert rel_inf(y.cpu(), ref.float().cpu()) < 1e-4
    assert rel_inf(out16.float().cpu(), ref.float().cpu()) < 1e-2
    assert torch.equal(out16, y.to(torch.bfloat16))


def test_bf16_kernel_grouped_gather():
    L, ops = _mods()
    E, D, F_, Mtok = 5, 128, 192, 333
    x = _rand(Mtok, D, seed=31).to(torch.bfloat16)
    w, b = _rand(E, F_, D, seed=32), _rand(E, F_, seed=33)
    counts = torch.tensor([0, 130, 1, 257, 61])
    off = torch.zeros(E + 1, dtype=torch.int32)
    off[1:] = counts.cumsum(0)
    tot = int(off[-1])
    gather = torch.randint(0, Mtok, (tot,), generator=torch.Generator().manual_seed(3), dtype=torch.int32)
    pw = ops.PackedWeight(w)
    out = torch.zeros(tot, F_, device="cuda")
    d = ops.gemm_desc(1)
    gd, od = gather.cuda(), off.cuda()
    d.A.p, d.A.ld, d.A.kind, d.A.gather = x.data_ptr(), D, L.OP_BF16_ROW, gd.data_ptr()
    d.W = pw.operand()
    d.W.bs1 = F_ * pw.Kp
    d.goff, d.ngroups = od.data_ptr(), E
    d.M, d.N, d.K = tot, F_, D
    d.bias, d.bias_bs = b.data_ptr(), F_
    d.C, d.ldc = out.data_ptr(), F_
    d.act = L.ACT_GELU
    ops.run_gemm(d)
    ref = torch.empty(tot, F_, dtype=torch.float64)
    wb = w.to(torch.bfloat16).cpu().double()
    for e in range(E):
        r = slice(int(off[e]), int(off[e + 1]))
        ref[r] = torch.nn.functional.gelu(x.cpu().double()[gather[r].long()] @ wb[e].T + b[e].cpu().double())
    assert rel_inf(out.cpu(), ref.float()) < 1e-4


@pytest.fixture
def force_256():
    L, _ = _mods()
    L.lib().mdm_set_gemm_variant(6)  # force the 256x256-tile kernel (gemm4.hip) where eligible
    yield
    L.lib().mdm_set_gemm_variant(0)


@pytest.mark.parametrize("M,N,K,act", [(512, 256, 64, 0), (12544, 512, 512, 1), (700, 768, 128, 2), (255, 1024, 192, 0), (6272, 1024, 1024, 0),
                                        (1000, 256, 4096, 1)])
def test_bf16_256_tile_kernel(force_256, M, N, K, act):
    L, ops = _mods()
    x, w, b = _rand(M, K, seed=41), _rand(N, K, seed=42), _rand(N, seed=43)
    xb = x.to(torch.bfloat16)
    pw = ops.PackedWeight(w)
    r1, r2, cs, rs = _rand(M, N, seed=44), _rand(M, N, seed=45), _rand(N, seed=46), _rand(M, seed=47)
    out16 = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    y = ops.linear(xb, pw, b, act=act, alpha=0.9, out_scale=0.5, colscale=cs, rowscale=rs, r1=r1, r1_scale=0.3, r2=r2,
                   precision=1, out16=out16)
    fn = {0: lambda v: v, 1: torch.nn.functional.gelu, 2: torch.nn.functional.silu}[act]
    ref = fn(0.9 * (xb.double() @ w.to(torch.bfloat16).double().T + b.double())) * 0.5
    ref = ref * cs.double()[None] * rs.double()[:, None] + 0.3 * r1.double() + r2.double()
    assert rel_inf(y.cpu(), ref.float().cpu()) < 1e-4
    assert torch.equal(out16, y.to(torch.bfloat16))


def test_bf16_256_tile_grouped_gather(force_256):
    L, ops = _mods()
    E, D, F_, Mtok = 5, 128, 256, 900
    x = _rand(Mtok, D, seed=51).to(torch.bfloat16)
    w, b = _rand(E, F_, D, seed=52), _rand(E, F_, seed=53)
    counts = torch.tensor([0, 300, 1, 513, 256])
    off = torch.zeros(E + 1, dtype=torch.int32)
    off[1:] = counts.cumsum(0)
    tot = int(off[-1])
    gather = torch.randint(0, Mtok, (tot,), generator=torch.Generator().manual_seed(3), dtype=torch.int32)
    pw = ops.PackedWeight(w)
    out = torch.zeros(tot, F_, device="cuda")
    rs = _rand(tot, seed=54)
    d = ops.gemm_desc(1)
    gd, od = gather.cuda(), off.cuda()
    d.A.p, d.A.ld, d.A.kind, d.A.gather = x.data_ptr(), D, L.OP_BF16_ROW, gd.data_ptr()
    d.W = pw.operand()
    d.W.bs1 = F_ * pw.Kp
    d.goff, d.ngroups = od.data_ptr(), E
    d.M, d.N, d.K = tot, F_, D
    d.bias, d.bias_bs = b.data_ptr(), F_
    d.rowscale = rs.data_ptr()
    d.C, d.ldc = out.data_ptr(), F_
    d.act = L.ACT_GELU
    ops.run_gemm(d)
    ref = torch.empty(tot, F_, dtype=torch.float64)
    wb = w.to(torch.bfloat16).cpu().double()
    for e in range(E):
        r = slice(int(off[e]), int(off[e + 1]))
        ref[r] = torch.nn.functional.gelu(x.cpu().double()[gather[r].long()] @ wb[e].T + b[e].cpu().double())
    ref = ref * rs.cpu().double()[:, None]
    assert rel_inf(out.cpu(), ref.float()) < 1e-4


@pytest.mark.parametrize("M,N,K,grouped", [(12544, 512, 512, False), (300, 263, 1024, False), (515, 1024, 512, True)])
def test_x3_dma_kernel_against_the_register_staged_one(M, N, K, grouped):
    """fp32-grade Linears run on the LDS-DMA staged bf16x3 kernel (csrc/gemm3.hip); knob 36 selects the register-staged
    kernel (csrc/gemm.hip).  Same split operands, different accumulation order: both within 2e-5 of fp64, and different bits."""
    L, ops = _mods()
    G = 3 if grouped else 1
    x, b = _rand(M, K, seed=1), _rand(G, N, seed=3)
    w = _rand(G, N, K, seed=2) * K ** -0.5
    pw = ops.PackedWeight(w if grouped else w[0])
    sizes = [200, 0, 315] if grouped else [M]
    goff = torch.tensor([0] + list(torch.tensor(sizes).cumsum(0)), dtype=torch.int32, device="cuda")
    g = torch.Generator(device="cpu").manual_seed(5)
    gather = torch.randint(0, M, (M,), generator=g, dtype=torch.int32).cuda() if grouped else None
    rs = _rand(M, seed=6).abs()
    r1 = _rand(M, N, seed=7)

    def run():
        d = ops.gemm_desc(3)
        d.A = ops.f32_operand(x, K)
        d.A.gather = L.ptr(gather)
        d.W = pw.operand()
        d.M, d.N, d.K = M, N, K
        out = torch.zeros(M, N, device="cuda")
        d.C, d.ldc = out.data_ptr(), N
        d.bias, d.act, d.rowscale = b.data_ptr(), L.ACT_GELU, rs.data_ptr()
        d.R1, d.ldr1, d.r1_scale = r1.data_ptr(), N, 0.5
        if grouped:
            d.goff, d.ngroups, d.W.bs1, d.bias_bs = goff.data_ptr(), G, N * pw.Kp, N
        ops.run_gemm(d)
        return out

    new = run()
    L.lib().mdm_set_gemm_variant(36)
    try:
        old = run()
    finally:
        L.lib().mdm_set_gemm_variant(0)
    xs = x[gather.long()] if grouped else x
    ref = torch.empty(M, N, dtype=torch.float64, device="cuda")
    o = 0
    for e, n in enumerate(sizes):
        ref[o:o + n] = torch.nn.functional.gelu(xs[o:o + n].double() @ w[e].double().T + b[e].double())
        o += n
    ref = ref * rs.double()[:, None] + 0.5 * r1.double()
    e_new, e_old = rel_inf(new.cpu(), ref.float().cpu()), rel_inf(old.cpu(), ref.float().cpu())
    print(f"x3 {M}x{N}x{K} grouped={grouped}: LDS-DMA kernel {e_new:.2e}, register-staged kernel {e_old:.2e}")
    assert e_new < 2e-5 and e_old < 2e-5


@pytest.mark.parametrize("M", [300, 64, 1])
def test_headnorm_epilogue_writes_normalised_hi_lo_planes(M):
    """MDM_ACT_HEADNORM (csrc/gemm3.hip): every 128-column slice of 0.1 * (x W^T + b) gets the LayerNorm over head_dim, the
    first `hn_l2_tiles` slices also the L2 normalisation (fast_attention.py:44-55), and the rows come out as bf16 hi / lo planes
    whose sum is the fp32 value to ~2^-17.  Reference: fp64 GEMM + torch layer_norm / normalize."""
    import ctypes as C
    import torch.nn.functional as F
    L, ops = pkg("_lib"), pkg("ops")
    D, H = 512, 4
    g = torch.Generator().manual_seed(5)
    x = (torch.rand(M, D, generator=g) * 2 - 1).cuda()
    w = ((torch.rand(3 * D, D, generator=g) * 2 - 1) * D ** -0.5).cuda()
    b = ((torch.rand(3 * D, generator=g) * 2 - 1) * 0.1).cuda()
    hw, hb = (1 + 0.2 * (torch.rand(128, generator=g) * 2 - 1)).cuda(), (0.1 * (torch.rand(128, generator=g) * 2 - 1)).cuda()
    pw = ops.PackedWeight(w)
    hi = torch.zeros(M, 3 * D, dtype=torch.bfloat16, device="cuda")
    lo = torch.zeros_like(hi)
    d = ops.gemm_desc(3)
    d.A = ops.f32_operand(x, x.stride(0))
    d.W = pw.operand()
    d.M, d.N, d.K = M, 3 * D, D
    d.bias, d.alpha, d.act = b.data_ptr(), 0.1, L.ACT_HEADNORM
    d.hn_w, d.hn_b, d.hn_l2_tiles = hw.data_ptr(), hb.data_ptr(), 2 * H
    d.C16, d.C16_lo, d.ldc = hi.data_ptr(), lo.data_ptr(), 3 * D
    ops.run_gemm(d)
    y = (0.1 * (x.double() @ w.double().T + b.double())).reshape(M, 3 * H, 128)
    y = F.layer_norm(y, (128,), hw.double(), hb.double(), 1e-5)
    y[:, :2 * H] = F.normalize(y[:, :2 * H], dim=-1)
    got = (hi.double() + lo.double()).reshape(M, 3 * H, 128)
    err = float((got - y).abs().max() / y.abs().max())
    print(f"M={M}: headnorm planes vs fp64 {err:.2e}")
    assert err < 3e-5
    assert float((hi.double().reshape(M, 3 * H, 128) - y).abs().max() / y.abs().max()) < 5e-3  # hi alone: one bf16 rounding
    d.C16_lo = None  # the lo plane is mandatory
    with pytest.raises(L.MdmError):
        ops.run_gemm(d)


@pytest.mark.parametrize("M", [300, 1])
def test_plane_outputs_and_head_softmax_epilogue(M):
    """csrc/gemm3.hip: (a) MDM_ACT_NONE with C16_lo: the result (with bias, alpha, residual) as bf16 hi / lo planes; (b)
    MDM_ACT_HEADSOFTMAX: softmax over every 128-column slice, as planes.  References in fp64."""
    L, ops = _mods()
    D = 512
    x, w, b = _rand(M, D, seed=1), _rand(D, D, seed=2) * D ** -0.5, _rand(D, seed=3)
    r1 = _rand(M, D, seed=4)
    pw = ops.PackedWeight(w)
    base = x.double() @ w.double().T + b.double()
    for act, ref in ((L.ACT_NONE, 0.25 * base + 0.5 * r1.double()),
                     (L.ACT_HEADSOFTMAX, torch.softmax(base.reshape(M, 4, 128), -1).reshape(M, D))):
        hi = torch.zeros(M, D, dtype=torch.bfloat16, device="cuda")
        lo = torch.zeros_like(hi)
        d = ops.gemm_desc(3)
        d.A = ops.f32_operand(x, D)
        d.W = pw.operand()
        d.M, d.N, d.K = M, D, D
        d.bias, d.act = b.data_ptr(), act
        if act == L.ACT_NONE:
            d.alpha = 0.25
            d.R1, d.ldr1, d.r1_scale = r1.data_ptr(), D, 0.5
        d.C16, d.C16_lo, d.ldc = hi.data_ptr(), lo.data_ptr(), D
        ops.run_gemm(d)
        err = float(((hi.double() + lo.double()) - ref).abs().max() / ref.abs().max())
        print(f"M={M} act={act}: planes vs fp64 {err:.2e}")
        assert err < 3e-5


@pytest.mark.parametrize("M,N,K,grouped", [(12544, 512, 512, False), (300, 1024, 512, False), (515, 512, 1024, True)])
def test_x3_gemm_on_pre_split_rows_is_bit_identical(M, N, K, grouped):
    """MDM_OP_X2_ROW activations (csrc/gemm3.hip AX2): a first GEMM writes its result pre-split (MdmGemmDesc.Cx2) beside the fp32
    copy, a second GEMM reads either form (dense, and grouped with a row gather): bit-identical results, and Cx2 holds exactly
    split_bf16 of the fp32 copy in the documented layout."""
    L, ops = _mods()
    x = _rand(M, 256, seed=1)
    w0 = _rand(K, 256, seed=2) * 256 ** -0.5
    G = 3 if grouped else 1
    w1 = _rand(G, N, K, seed=3) * K ** -0.5
    pw0, pw1 = ops.PackedWeight(w0), ops.PackedWeight(w1 if grouped else w1[0])
    mid = torch.zeros(M, K, device="cuda")
    midx2 = torch.zeros(M, 2 * K, dtype=torch.bfloat16, device="cuda")
    d = ops.gemm_desc(3)
    d.A = ops.f32_operand(x, 256)
    d.W = pw0.operand()
    d.M, d.N, d.K = M, K, 256
    d.C, d.Cx2, d.ldc, d.act = mid.data_ptr(), midx2.data_ptr(), K, L.ACT_GELU
    ops.run_gemm(d)
    hi = mid.to(torch.bfloat16)
    lo = (mid - hi.float()).to(torch.bfloat16)
    want = torch.stack([hi.reshape(M, K // 32, 32), lo.reshape(M, K // 32, 32)], 2).reshape(M, 2 * K)
    assert torch.equal(midx2.view(torch.int16), want.view(torch.int16))
    sizes = [200, 0, 315] if grouped else [M]
    goff = torch.tensor([0] + list(torch.tensor(sizes).cumsum(0)), dtype=torch.int32, device="cuda")
    g = torch.Generator(device="cpu").manual_seed(5)
    gather = torch.randint(0, M, (M,), generator=g, dtype=torch.int32).cuda() if grouped else None
    outs = []
    for kind, buf in ((L.OP_F32_ROW, mid), (L.OP_X2_ROW, midx2)):
        e = ops.gemm_desc(3)
        e.A = ops.f32_operand(buf, K)  # ld in 4-byte units in both forms
        e.A.kind = kind
        e.A.gather = L.ptr(gather)
        e.W = pw1.operand()
        e.M, e.N, e.K = sum(sizes), N, K
        out = torch.zeros(M, N, device="cuda")
        e.C, e.ldc = out.data_ptr(), N
        if grouped:
            e.goff, e.ngroups, e.W.bs1 = goff.data_ptr(), G, N * pw1.Kp
        ops.run_gemm(e)
        outs.append(out)
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("fmt", ["bf16", "f16"])
@pytest.mark.parametrize("M,N,K,act", [(12544, 1024, 1024, 1), (6272, 1024, 1024, 0), (3136, 1024, 1024, 0), (1568, 1024, 1024, 1),
                                       (1000, 1024, 1024, 0), (50, 512, 1024, 1), (1, 256, 512, 0), (12544, 512, 512, 0),
                                       (777, 3072, 1024, 0), (12544, 4096, 1024, 1)])
def test_streamed_weight_gemm(M, N, K, act, fmt):
    """csrc/gemm_stream.hip (MdmGemmDesc.w_stream): the weight as a fragment stream global -> registers, the 16-bit rows through LDS
    in K slices; every tile shape the launcher picks (112 / 64 / 32 rows, 512 / 256 columns), ragged last row tiles, both 16-bit
    formats, the whole epilogue.  Against fp64 of the same rounded operands, and against the tile kernel (knob 63) it replaces."""
    L, ops = _mods()
    dt = torch.float16 if fmt == "f16" else torch.bfloat16
    x, w, b = _rand(M, K, seed=41), _rand(N, K, seed=42) * K ** -0.5 * 4, _rand(N, seed=43)
    xb = x.to(dt)
    pw = ops.PackedWeight(w, fmt=fmt)
    ws = ops.gemm_stream1_pack(w, dt)
    assert ws is not None and ws.numel() >= N * K + 16 * 512
    r1, cs = _rand(M, N, seed=44), _rand(N, seed=45)
    kw = dict(act=L.ACT_GELU if act else L.ACT_NONE, alpha=0.9, out_scale=0.5, colscale=cs, r1=r1, r1_scale=0.3, precision=1)
    o16s, o16t = torch.empty(M, N, dtype=dt, device="cuda"), torch.empty(M, N, dtype=dt, device="cuda")
    L.lib().mdm_set_gemm_variant(68)  # the streamed kernel wherever it is eligible (by default: only where it measured faster)
    try:
        y = ops.linear(xb, pw, b, out16=o16s, w_stream=ws, **kw)
        y2 = ops.linear(xb, pw, None, precision=1, w_stream=ws)
    finally:
        L.lib().mdm_set_gemm_variant(0)
    L.lib().mdm_set_gemm_variant(63)
    try:
        yt = ops.linear(xb, pw, b, out16=o16t, w_stream=ws, **kw)
    finally:
        L.lib().mdm_set_gemm_variant(0)
    pre = 0.9 * (xb.double() @ w.to(dt).double().T + b.double())
    ref = (torch.nn.functional.gelu(pre) if act else pre) * 0.5 * cs.double()[None] + 0.3 * r1.double()
    e_ref, e_tile = rel_inf(y.cpu(), ref.float().cpu()), rel_inf(y.cpu(), yt.cpu())
    print(f"M={M} N={N} K={K} act={act} {fmt}: vs fp64 {e_ref:.2e}, vs the tile kernel {e_tile:.2e} (equal: {torch.equal(y, yt)})")
    assert e_ref < 1e-4 and e_tile < 2e-6
    assert torch.equal(o16s, y.to(dt))
    # outputs one at a time, no residual / scales: the optional pointers of the epilogue
    assert rel_inf(y2.cpu(), (xb.double() @ w.to(dt).double().T).float().cpu()) < 1e-4


def test_streamed_weight_gemm_is_only_taken_where_it_applies():
    """A launch the streamed kernel does not cover (row scales, K outside {512, 1024}, fp32 rows) runs on the tile kernels as if no
    stream had been given."""
    L, ops = _mods()
    M, N, K = 300, 512, 1024
    x, w = _rand(M, K, seed=51), _rand(N, K, seed=52)
    pw, ws = ops.PackedWeight(w), ops.gemm_stream1_pack(w, torch.bfloat16)
    rs = _rand(M, seed=53)
    a = ops.linear(x.to(torch.bfloat16), pw, None, rowscale=rs, precision=1, w_stream=ws)
    b = ops.linear(x.to(torch.bfloat16), pw, None, rowscale=rs, precision=1)
    assert torch.equal(a, b)
    a = ops.linear(x, pw, None, precision=3, w_stream=ws)
    b = ops.linear(x, pw, None, precision=3)
    assert torch.equal(a, b)
    assert ops.gemm_stream1_pack(_rand(200, 512, seed=54), torch.bfloat16) is None


def _x2_rows(x):
    """fp32 [M, K] -> MDM_OP_X2_ROW rows [M, 2 K] bf16 (per 32 columns: 32 hi then 32 lo = rn(x - hi))."""
    M, K = x.shape
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    return torch.stack([hi.reshape(M, K // 32, 32), lo.reshape(M, K // 32, 32)], 2).reshape(M, 2 * K).contiguous()


@pytest.mark.parametrize("sizes,N,K,act,gathered", [([3136] * 16, 1024, 512, 1, True), ([3136] * 16, 512, 1024, 0, False),
                                                    ([200, 0, 315, 112, 1, 113], 1024, 512, 1, True),
                                                    ([200, 0, 315, 112, 1, 113], 512, 1024, 0, False), ([777], 512, 512, 0, False),
                                                    ([5], 1024, 1024, 1, True)])
def test_streamed_weight_x3_gemm_is_bit_identical_to_the_tile_kernel(sizes, N, K, act, gathered):
    """csrc/gemm_stream3.hip: pre-split rows x a (hi, lo) fragment-pair stream, dense and grouped (empty, 1-row and ragged groups),
    with a row gather, GELU -> pre-split rows out (the expert W1 launch) and row scale -> fp32 out (W2).  Same products in the same
    order as the 128 x 128 tile kernel of csrc/gemm3.hip (knob 69): every output bit must agree; and against fp64."""
    L, ops = _mods()
    G, M = len(sizes), sum(sizes)
    S = max(M // 3, 8)  # source rows the gather draws from
    x = _rand(S if gathered else M, K, seed=61)
    w, b = _rand(G, N, K, seed=62) * K ** -0.5, _rand(G, N, seed=63)
    rs = _rand(M, seed=64).abs() + 0.1
    xx2 = _x2_rows(x)
    pw = ops.PackedWeight(w if G > 1 else w[0])
    ws = ops.gemm_stream3x_pack(w)
    assert ws is not None
    goff = torch.tensor([0] + list(torch.tensor(sizes).cumsum(0)), dtype=torch.int32, device="cuda")
    gather = torch.randint(0, S, (M,), generator=torch.Generator().manual_seed(7), dtype=torch.int32).cuda() if gathered else None
    outs = []
    for knob in (70, 69):  # 70: the streamed kernel wherever it is eligible (by default only where it measured faster)
        d = ops.gemm_desc(3)
        d.A = ops.f32_operand(xx2.view(torch.float32), K)  # ld in 4-byte units
        d.A.kind, d.A.gather = L.OP_X2_ROW, L.ptr(gather)
        d.W = pw.operand()
        d.w_stream = ws.data_ptr()
        d.M, d.N, d.K = M, N, K
        d.bias, d.act = b.data_ptr(), (L.ACT_GELU if act else L.ACT_NONE)
        if G > 1:
            d.goff, d.ngroups, d.W.bs1, d.bias_bs = goff.data_ptr(), G, N * pw.Kp, N
            d.w_stream_gs = L.lib().mdm_gemm_stream3x_group_elems(C.c_int32(N), C.c_int32(K))
        out, ox2 = torch.zeros(M, N, device="cuda"), torch.zeros(M, 2 * N, dtype=torch.bfloat16, device="cuda")
        d.C, d.ldc = out.data_ptr(), N
        if act:
            d.Cx2 = ox2.data_ptr()
        else:
            d.rowscale = rs.data_ptr()
        L.lib().mdm_set_gemm_variant(knob)
        try:
            ops.run_gemm(d)
        finally:
            L.lib().mdm_set_gemm_variant(0)
        outs.append((out, ox2))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1].view(torch.int16), outs[1][1].view(torch.int16))
    xs = (x[gather.long()] if gathered else x).double()
    ref = torch.empty(M, N, dtype=torch.float64, device="cuda")
    for gi in range(G):
        r = slice(int(goff[gi]), int(goff[gi + 1]))
        ref[r] = xs[r] @ w[gi].double().T + b[gi].double()
    ref = torch.nn.functional.gelu(ref) if act else ref * rs.double()[:, None]
    assert rel_inf(outs[0][0].cpu(), ref.float().cpu()) < 2e-5
    if act:
        assert torch.equal(outs[0][1].view(torch.int16), _x2_rows(outs[0][0]).view(torch.int16))
