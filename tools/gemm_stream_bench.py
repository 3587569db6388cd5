"""Streamed-weight GEMM (csrc/gemm_stream.hip) against the 128 x 128 tile kernel (knob 63) at the big model's shapes, interleaved in one
process: 16-bit rows, fp32 + 16-bit outputs, bias + residual epilogue; knobs 64..67 force the tile shape (rows x columns)."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("motiondiffusion-moe_amd.ops")
L = importlib.import_module("motiondiffusion-moe_amd._lib")


def main():
    dev = "cuda"
    light = len(sys.argv) > 1 and sys.argv[1] == "light"  # 16-bit output only, no residual: the launch without its fp32 epilogue traffic
    shapes = [(12544, 1024, 1024), (6272, 1024, 1024), (3136, 1024, 1024), (1568, 1024, 1024), (12544, 3072, 1024), (12544, 4096, 1024),
              (6272, 3072, 1024), (12544, 512, 512), (6272, 512, 512)]
    variants = [(63, "tile kernel"), (68, "streamed (auto shape)"), (64, "112 x 512"), (65, "64 x 512"), (66, "64 x 256"), (67, "32 x 256")]
    for M, N, K in shapes:
        torch.manual_seed(0)
        x = torch.randn(M, K, device=dev).to(torch.bfloat16)
        w = torch.randn(N, K, device=dev) * K ** -0.5
        b, r1 = torch.randn(N, device=dev), torch.randn(M, N, device=dev)
        pw, ws = ops.PackedWeight(w), ops.gemm_stream1_pack(w, torch.bfloat16)
        out, o16 = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev, dtype=torch.bfloat16)

        def run(v):
            L.lib().mdm_set_gemm_variant(v)
            if light:
                d = ops.gemm_desc(1)
                d.A.p, d.A.ld, d.A.kind = x.data_ptr(), K, L.OP_BF16_ROW
                d.W, d.w_stream = pw.operand(), ws.data_ptr()
                d.M, d.N, d.K, d.bias, d.C16, d.ldc, d.h16 = M, N, K, b.data_ptr(), o16.data_ptr(), N, L.H16_BF16
                ops.run_gemm(d)
            else:
                ops.linear(x, pw, b, r1=r1, precision=1, out=out, out16=o16, w_stream=ws)
            L.lib().mdm_set_gemm_variant(0)

        res = {v: [] for v, _ in variants}
        for _ in range(5):
            for v, _n in variants:
                run(v)
        for _rnd in range(5):
            for v, _n in variants:
                torch.cuda.synchronize()
                a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(10):
                    run(v)
                e.record()
                torch.cuda.synchronize()
                res[v].append(a.elapsed_time(e) / 10 * 1e3)
        flop = 2.0 * M * N * K
        med = {v: sorted(r)[len(r) // 2] for v, r in res.items()}
        print(f"M={M} N={N} K={K}: " + "  ".join(f"{n} {med[v]:.1f} us ({flop / med[v] / 1e6:.0f} TF)" for v, n in variants), flush=True)


if __name__ == "__main__":
    main()
