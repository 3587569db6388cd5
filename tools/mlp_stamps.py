"""Where a tile of the streamed-weight fused MLP spends its cycles: the diagnostic build (knob 49) sums s_memtime
differences per phase in wave 0 of every workgroup (csrc/mlp_stream.hip, XSTAMP).  Shares, not run time.
Runs on the DIAGNOSTIC library (python motiondiffusion-moe_amd/build.py --diag): the product library has no stamped build."""
import ctypes
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MDM_LIB", os.path.join(ROOT, "motiondiffusion-moe_amd", "libmdm_hip_diag.so"))
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
        st = torch.zeros(8, dtype=torch.int64, device=dev)  # the stamped build's counters: a buffer of their own
        assert L.lib().mdm_diag_build() == 1, "build the diagnostic library first: python motiondiffusion-moe_amd/build.py --diag"
        L.check(L.lib().mdm_diag_mlp_counters(ctypes.c_void_p(st.data_ptr())))

        def run(v):
            L.check(L.lib().mdm_set_gemm_variant(v))
            ops.fused_mlp(x16, pw1, b1, pw2, b2, gather=gather, goff=goff, rowscale=rs, rows=M, out16=out16, wstream=ws, only16=True)
            L.lib().mdm_set_gemm_variant(0)

        for _ in range(50):
            run(0)
        st.zero_()
        n = 10
        for _ in range(n):
            run(49)
        torch.cuda.synchronize()
        L.lib().mdm_diag_mlp_counters(None)
        v = st.cpu().tolist()
        tiles = v[7] / n
        tot = sum(v[:7])
        print(f"M={M}: {tiles:.0f} tiles per launch, {tot / v[7]:.0f} cycles per tile (wave 0, stamped build)")
        for k in range(7):
            print(f"    {NAMES[k]:36s} {v[k] / v[7]:9.0f} cycles  {100.0 * v[k] / tot:5.1f} %")


if __name__ == "__main__":
    main()
