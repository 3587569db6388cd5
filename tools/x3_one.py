"""One fp32-grade GEMM shape, 20 launches (for counter passes): python tools/x3_one.py M N K"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ops = importlib.import_module("motiondiffusion-moe_amd.ops")
M, N, K = (int(v) for v in sys.argv[1:4])
x = torch.randn(M, K, device="cuda")
w = torch.randn(N, K, device="cuda") * K ** -0.5
pw = ops.PackedWeight(w)
out = torch.empty(M, N, device="cuda")
for _ in range(20):
    ops.linear(x, pw, None, precision=3, out=out)
torch.cuda.synchronize()
