"""Same-process, interleaved timing of the two fused expert-MLP kernels (csrc/mlp.hip vs csrc/mlp2.hip) at the bench shapes:
50176 / 25088 routed rows over 16 expert slabs, D = 512, F = 1024, gathered rows, 16-bit output."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("motiondiffusion-moe_amd.ops")
L = importlib.import_module("motiondiffusion-moe_amd._lib")


def main():
    dev, D, F, G = "cuda", 512, 1024, 16
    dtype = torch.float16 if "--bf16" not in sys.argv else torch.bfloat16
    fmt = "f16" if dtype == torch.float16 else "bf16"
    for M in (50176, 25088):
        torch.manual_seed(0)
        S = M // 4
        x16 = torch.randn(2 * S, D, device=dev).to(dtype)
        w1 = torch.randn(G, F, D, device=dev) * D ** -0.5
        w2 = torch.randn(G, D, F, device=dev) * F ** -0.5
        b1, b2 = torch.randn(G, F, device=dev) * 0.1, torch.randn(G, D, device=dev) * 0.1
        pw1, pw2 = ops.PackedWeight(w1, fmt=fmt), ops.PackedWeight(w2, fmt=fmt)
        frag = ops.mlp_fragment_major(w1, w2, dtype)
        gather = torch.randint(0, 2 * S, (M,), device=dev, dtype=torch.int32)
        goff = (torch.arange(G + 1, device=dev, dtype=torch.int64) * M // G).to(torch.int32)
        rs = torch.rand(M, device=dev)
        out = torch.empty(M, D, device=dev)
        out16 = torch.empty(M, D, device=dev, dtype=dtype)

        def run(variant):
            L.lib().mdm_set_gemm_variant(variant)
            ops.fused_mlp(x16, pw1, b1, pw2, b2, gather=gather, goff=goff, rowscale=rs, rows=M, out=out, out16=out16, frag=frag)
            L.lib().mdm_set_gemm_variant(0)

        variants = [(35, 'gen2'), (0, 'gen1'), (41, 'gen2 no GELU'), (42, 'gen2 no DMA'), (43, 'gen2 no frag reads'), (44, 'gen2 MFMA only')]
        res = {v: [] for v, _ in variants}
        for _ in range(30):  # warm clocks
            run(0), run(34)
        for rnd in range(8):
            for v, _ in variants:
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(10):
                    run(v)
                b.record()
                torch.cuda.synchronize()
                res[v].append(a.elapsed_time(b) / 10 * 1e3)
        flop = 4.0 * M * D * F
        for v, name in variants:
            t = sorted(res[v])
            med, mn = t[len(t) // 2], t[0]
            print(f"M={M} {fmt} {name}: median {med:.1f} us  min {mn:.1f} us  -> {flop / med / 1e6:.0f} TFLOP/s "
                  f"({flop / med / 1e6 / 2500:.3f} of 2.5 PF)", flush=True)


if __name__ == "__main__":
    main()
