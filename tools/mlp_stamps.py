"""Where a tile of the streamed-weight fused MLP spends its cycles: the diagnostic build (knob 49) sums s_memtime
differences per phase in wave 0 of every workgroup (csrc/mlp_stream.hip, XSTAMP).  Shares, not run time."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("motiondiffusion-moe_amd.ops")
L = importlib.import_module("motiondiffusion-moe_amd._lib")
NAMES = ["tile lookup", "X tile -> LDS, ring fill, barrier", "phase 1 (X . W1)", "GELU", "hidden image: barriers + writes",
         "phase 2 (hidden . W2)", "epilogue"]


def main():
    dev, D, F, G = "cuda", 512, 1024, 16
    for M in (50176, 25088):
        torch.manual_seed(0)
        S = 12544
        x16 = torch.randn(S, D, device=dev).to(torch.float16)
        w1 = torch.randn(G, F, D, device=dev) * D ** -0.5
        w2 = torch.randn(G, D, F, device=dev) * F ** -0.5
        b1, b2 = torch.randn(G, F, device=dev) * 0.1, torch.randn(G, D, device=dev) * 0.1
        pw1, pw2 = ops.PackedWeight(w1, fmt="f16"), ops.PackedWeight(w2, fmt="f16")
        ws = ops.mlp_stream_pack(w1, w2, torch.float16)
        out16 = torch.empty(M, D, device=dev, dtype=torch.float16)
        out = torch.empty(M, D, device=dev)
        gather = torch.randint(0, S, (M,), device=dev, dtype=torch.int32)
        goff = (torch.arange(G + 1, dtype=torch.int64) * M // G).to(torch.int32).to(dev)
        rs = torch.rand(M, device=dev)
        st = torch.zeros(8, dtype=torch.int64, device=dev)
        fake_r2 = st.view(torch.float32).reshape(1, 16)  # the diagnostic build takes the counters through the R2 pointer

        def run(v, r2=None):
            L.lib().mdm_set_gemm_variant(v)
            if r2 is None:
                ops.fused_mlp(x16, pw1, b1, pw2, b2, gather=gather, goff=goff, rowscale=rs, rows=M, out16=out16, wstream=ws, only16=True)
            else:  # the stamped build writes 16-bit outputs only as well (it takes its counters through R2 and ignores C)
                ops.fused_mlp(x16, pw1, b1, pw2, b2, gather=gather, goff=goff, rowscale=rs, rows=M, out16=out16, wstream=ws, r2=r2, only16=True)
            L.lib().mdm_set_gemm_variant(0)

        for _ in range(50):
            run(0)
        st.zero_()
        n = 10
        for _ in range(n):
            run(49, fake_r2)
        torch.cuda.synchronize()
        v = st.cpu().tolist()
        tiles = v[7] / n
        tot = sum(v[:7])
        print(f"M={M}: {tiles:.0f} tiles per launch, {tot / v[7]:.0f} cycles per tile (wave 0, stamped build)")
        for k in range(7):
            print(f"    {NAMES[k]:36s} {v[k] / v[7]:9.0f} cycles  {100.0 * v[k] / tot:5.1f} %")


if __name__ == "__main__":
    main()
