"""Interleaved same-process timing of the fp32-grade (bf16x3) GEMM variants at the step's shapes (knobs 0 / 36 / 37 / 38 / 39)."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("motiondiffusion-moe_amd.ops")
L = importlib.import_module("motiondiffusion-moe_amd._lib")


def main():
    dev = "cuda"
    shapes = [("DxD full", 12544, 512, 512, 0), ("DxD half", 6272, 512, 512, 0), ("qkv full", 12544, 1536, 512, 0),
              ("ffn1 full", 12544, 2048, 512, 0), ("ffn2 full", 12544, 512, 2048, 0), ("expert W1", 50176, 1024, 512, 16),
              ("expert W2", 50176, 512, 1024, 16)]
    variants = [(0, "default"), (36, "register-staged"), (37, "<128,128,3>"), (38, "<128,128,4>"), (39, "<64,128,3>")]
    for name, M, N, K, G in shapes:
        torch.manual_seed(0)
        S = M // 4 if G else M
        x = torch.randn(S, K, device=dev)
        w = torch.randn(*((G,) if G else ()), N, K, device=dev) * K ** -0.5
        pw = ops.PackedWeight(w)
        out = torch.empty(M, N, device=dev)
        gather = torch.randint(0, S, (M,), device=dev, dtype=torch.int32) if G else None
        goff = (torch.arange(G + 1, device=dev, dtype=torch.int64) * M // G).to(torch.int32) if G else None

        def run(v):
            L.lib().mdm_set_gemm_variant(v)
            d = ops.gemm_desc(3)
            d.A = ops.f32_operand(x, K)
            d.A.gather = L.ptr(gather)
            d.W = pw.operand()
            d.M, d.N, d.K = M, N, K
            d.C, d.ldc = out.data_ptr(), N
            if G:
                d.goff, d.ngroups, d.W.bs1 = goff.data_ptr(), G, N * pw.Kp
            ops.run_gemm(d)
            L.lib().mdm_set_gemm_variant(0)

        res = {v: [] for v, _ in variants}
        for _ in range(10):
            for v, _n in variants:
                run(v)
        for rnd in range(6):
            for v, _n in variants:
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(10):
                    run(v)
                b.record()
                torch.cuda.synchronize()
                res[v].append(a.elapsed_time(b) / 10 * 1e3)
        flop = 2.0 * M * N * K
        print(name, f"M={M} N={N} K={K}:", "  ".join(f"{n} {sorted(res[v])[len(res[v]) // 2]:.1f} us" for v, n in variants),
              f"  (1x flop at best: {flop / min(min(r) for r in res.values()) / 1e6:.0f} TF)", flush=True)


if __name__ == "__main__":
    main()
