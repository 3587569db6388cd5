"""Microbench of the fused GEMMs on denoiser shapes, timed inside a captured hipGraph (no host launch overhead)."""
import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ops = importlib.import_module("motiondiffusion-moe_amd.ops")
L = importlib.import_module("motiondiffusion-moe_amd._lib")

def bench(M, N, K, precision, act=0, a16=False, reps=20, out16=False):
    x = torch.randn(M, K, device="cuda"); x = x.to(torch.bfloat16) if a16 else x
    w = torch.randn(N, K, device="cuda") * K ** -0.5
    b = torch.randn(N, device="cuda"); pw = ops.PackedWeight(w); out = torch.empty(M, N, device="cuda")
    o16 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16) if out16 else None
    run = lambda: ops.linear(x, pw, b, act=act, precision=precision, out=out, out16=o16)
    for _ in range(3): run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): run()
    g.replay(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (3 * reps)
    return us, 2.0 * M * N * K / us / 1e6

if __name__ == "__main__":
    shapes = [(12544, 512, 512), (6272, 512, 512), (12544, 1536, 512), (12544, 2048, 512), (12544, 512, 2048),
              (50176, 1024, 512), (50176, 512, 1024), (12544, 1024, 1024), (4096, 4096, 4096)]
    print("v1 fp32-A kernel:")
    for (M, N, K) in shapes[:4]:
        us, tf = bench(M, N, K, 1, act=1)
        print(f"  M={M:6d} N={N:5d} K={K:5d}: {us:8.1f} us {tf:7.1f} TF")
    for v, name in [(1, "BK64 x2"), (2, "BK64 x3"), (3, "BK32 x3"), (4, "BK32 x4"), (5, "BK64 x4")]:
        L.lib().mdm_set_gemm_variant(v)
        print(f"bf16 glds kernel variant {v} ({name}):")
        for (M, N, K) in shapes:
            us, tf = bench(M, N, K, 1, act=1, a16=True)
            print(f"  M={M:6d} N={N:5d} K={K:5d}: {us:8.1f} us {tf:7.1f} TF")
