"""Same-box timing of the fused MLP kernel (mlp.hip) against the two-GEMM chain it replaces (gemm2.hip), at the
bench shapes (B=32 cfg-batched -> 64 x 196 = 12544 rows, D=512)."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("motiondiffusion-moe_amd.ops")
L = importlib.import_module("motiondiffusion-moe_amd._lib")


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    dev = "cuda"
    big = len(sys.argv) > 1 and sys.argv[1] == "big"   # the big model's widths (D = 1024, F = 2048)
    D = 1024 if big else 512
    cases = ([("moe_high", 50176, 2048, 16), ("moe_low", 25088, 2048, 16)] if big else
             [("moe_high", 50176, 1024, 16), ("moe_low", 25088, 1024, 16), ("ffn_high", 12544, 2048, 0),
              ("ffn_low", 6272, 2048, 0), ("proj_high", 12544, 512, 0), ("proj_low", 6272, 512, 0)])
    for name, M, F, G in cases:
        torch.manual_seed(0)
        S = 12544 if G else M
        x16 = torch.randn(S, D, device=dev).to(torch.bfloat16)
        lead = (G,) if G else ()
        w1 = torch.randn(*lead, F, D, device=dev) * D ** -0.5
        w2 = torch.randn(*lead, D, F, device=dev) * F ** -0.5
        b1 = torch.randn(*lead, F, device=dev) * 0.1
        b2 = torch.randn(*lead, D, device=dev) * 0.1
        pw1, pw2 = ops.PackedWeight(w1, with_lo=False), ops.PackedWeight(w2, with_lo=False)
        out = torch.empty(M, D, device=dev)
        hid = torch.empty(M, F, device=dev, dtype=torch.bfloat16)
        gather = goff = rs = None
        if G:
            gather = torch.randint(0, S, (M,), device=dev, dtype=torch.int32)
            goff = torch.arange(G + 1, device=dev, dtype=torch.int32) * (M // G)
            rs = torch.rand(M, device=dev)

        ws = ops.mlp_stream_pack(w1, w2, torch.bfloat16)

        def fused():
            if big:  # the LDS-staged kernel has no Dout = 1024 form
                return
            ops.fused_mlp(x16, pw1, b1, pw2, b2, gather=gather, goff=goff, rowscale=rs, rows=M, out=out)

        def stream():
            ops.fused_mlp(x16, pw1, b1, pw2, b2, gather=gather, goff=goff, rowscale=rs, rows=M, out=out, wstream=ws)

        def chain():
            d = ops.gemm_desc(1)
            d.A.p, d.A.ld, d.A.kind = x16.data_ptr(), D, L.OP_BF16_ROW
            d.A.gather = L.ptr(gather)
            d.W = pw1.operand()
            d.W.bs1 = F * pw1.Kp
            d.M, d.N, d.K = M, F, D
            d.C16, d.ldc = hid.data_ptr(), F
            d.bias, d.bias_bs, d.act = b1.data_ptr(), F, L.ACT_GELU
            if G:
                d.goff, d.ngroups = goff.data_ptr(), G
            ops.run_gemm(d)
            d = ops.gemm_desc(1)
            d.A.p, d.A.ld, d.A.kind = hid.data_ptr(), F, L.OP_BF16_ROW
            d.W = pw2.operand()
            d.W.bs1 = D * pw2.Kp
            d.M, d.N, d.K = M, D, F
            d.C, d.ldc = out.data_ptr(), D
            d.bias, d.bias_bs = b2.data_ptr(), D
            d.rowscale = L.ptr(rs)
            if G:
                d.goff, d.ngroups = goff.data_ptr(), G
            ops.run_gemm(d)

        fused()
        y1 = out.clone()
        stream()
        y2 = out.clone()
        chain()
        if big:
            y1 = out.clone()
        err = ((y1 - out).abs().max() / out.abs().max()).item()
        err2 = ((y2 - y1).abs().max() / y1.abs().max()).item()
        for _ in range(200):  # settle the clocks before comparing variants
            stream()
        torch.cuda.synchronize()
        rounds = [(timeit(fused), timeit(stream), timeit(chain)) for _ in range(3)]  # alternate the variants
        tf, ts, tc = (sorted(r[i] for r in rounds)[1] for i in range(3))
        fl = 4.0 * M * D * F
        print(f"{name:10s} M={M:6d} F={F:5d}  stream {ts:7.1f} us ({fl / ts / 1e6:6.0f} TF)   lds {tf:7.1f} us ({fl / tf / 1e6:6.0f} TF)   "
              f"chain {tc:7.1f} us ({fl / tc / 1e6:6.0f} TF)   lds-chain {err:.1e} stream-lds {err2:.1e}", flush=True)


if __name__ == "__main__":
    main()
