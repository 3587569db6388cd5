"""The fp32-grade expert GEMMs on the streamed-weight bf16x3 kernel (csrc/gemm_stream3.hip) against the 128 x 128 tile kernel
(knob 69), interleaved in one process: grouped (16 balanced groups), gathered pre-split rows, W1 + GELU -> pre-split rows and
W2 * row scale -> fp32, at the full (50176 routed rows) and the half time scale."""
import ctypes as C
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("motiondiffusion-moe_amd.ops")
L = importlib.import_module("motiondiffusion-moe_amd._lib")


def x2_rows(x):
    M, K = x.shape
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    return torch.stack([hi.reshape(M, K // 32, 32), lo.reshape(M, K // 32, 32)], 2).reshape(M, 2 * K).contiguous()


def main():
    dev, G = "cuda", 16
    for name, M, N, K, act in (("W1 full", 50176, 1024, 512, 1), ("W2 full", 50176, 512, 1024, 0), ("W1 half", 25088, 1024, 512, 1),
                               ("W2 half", 25088, 512, 1024, 0)):
        torch.manual_seed(0)
        S = M // 4
        x = x2_rows(torch.randn(S if act else M, K, device=dev))
        w = torch.randn(G, N, K, device=dev) * K ** -0.5
        b, rs = torch.randn(G, N, device=dev), torch.rand(M, device=dev)
        pw, ws = ops.PackedWeight(w), ops.gemm_stream3x_pack(w)
        goff = (torch.arange(G + 1, device=dev, dtype=torch.int64) * M // G).to(torch.int32)
        gather = torch.randint(0, S, (M,), device=dev, dtype=torch.int32) if act else None
        out, ox2 = torch.empty(M, N, device=dev), torch.empty(M, 2 * N, dtype=torch.bfloat16, device=dev)

        def run(v):
            d = ops.gemm_desc(3)
            d.A = ops.f32_operand(x.view(torch.float32), K)
            d.A.kind, d.A.gather = L.OP_X2_ROW, L.ptr(gather)
            d.W, d.w_stream = pw.operand(), ws.data_ptr()
            d.M, d.N, d.K, d.bias, d.bias_bs, d.act = M, N, K, b.data_ptr(), N, (L.ACT_GELU if act else L.ACT_NONE)
            d.goff, d.ngroups, d.W.bs1 = goff.data_ptr(), G, N * pw.Kp
            d.w_stream_gs = L.lib().mdm_gemm_stream3x_group_elems(C.c_int32(N), C.c_int32(K))
            d.ldc = N
            if act:
                d.Cx2 = ox2.data_ptr()
            else:
                d.C, d.rowscale = out.data_ptr(), rs.data_ptr()
            L.lib().mdm_set_gemm_variant(v)
            ops.run_gemm(d)
            L.lib().mdm_set_gemm_variant(0)

        res = {70: [], 69: []}
        for _ in range(5):
            run(70), run(69)
        for _rnd in range(5):
            for v in (69, 70):
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
        print(f"{name} M={M} N={N} K={K}: tile kernel {med[69]:.1f} us ({flop / med[69] / 1e6:.0f} TF)  streamed {med[70]:.1f} us "
              f"({flop / med[70] / 1e6:.0f} TF = {flop / med[70] / 1e6 / 833.3:.3f} of the bf16x3 peak)", flush=True)


if __name__ == "__main__":
    main()
