"""Microbench of the fused GEMM on denoiser shapes (run on the GPU box)."""
import importlib, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ops = importlib.import_module("motiondiffusion-moe_amd.ops")
L = importlib.import_module("motiondiffusion-moe_amd._lib")

def bench(M, N, K, precision, act=0, iters=50):
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * K ** -0.5
    b = torch.randn(N, device="cuda"); pw = ops.PackedWeight(w); out = torch.empty(M, N, device="cuda")
    for _ in range(5): ops.linear(x, pw, b, act=act, precision=precision, out=out)
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): ops.linear(x, pw, b, act=act, precision=precision, out=out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    tf = 2.0 * M * N * K / us / 1e6
    print(f"M={M:6d} N={N:5d} K={K:5d} prec={precision} act={act}: {us:8.1f} us  {tf:7.1f} TFLOP/s (x{3 if precision==3 else 1} MFMA)")

if __name__ == "__main__":
    for prec in (1, 3):
        for (M, N, K) in [(12544, 512, 512), (12544, 1536, 512), (12544, 2048, 512), (12544, 512, 2048), (6272, 512, 512),
                          (12544, 1024, 1024), (12544, 4096, 1024), (8192, 8192, 8192) if prec == 1 else (4096, 4096, 4096)]:
            bench(M, N, K, prec, act=1)
