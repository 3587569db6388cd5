"""How much of the fused expert-MLP launch is expert-weight L2 misses?  Same kernel, same rows (50176 gathered rows of 12544),
16 groups -- once with 16 distinct expert weight sets (33.5 MB, the real case) and once with group stride 0: all groups read
ONE 2-MB weight set (always L2-resident in every XCD).  Also: no row gather (contiguous X)."""
import ctypes as C, importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from mlp_bench import timeit
ops = importlib.import_module("motiondiffusion-moe_amd.ops")
L = importlib.import_module("motiondiffusion-moe_amd._lib")
dev, D, F, G, M, S = "cuda", 512, 1024, 16, 50176, 12544
torch.manual_seed(0)
x16 = torch.randn(S, D, device=dev).to(torch.float16)
xbig = torch.randn(M, D, device=dev).to(torch.float16)
gather = torch.randint(0, S, (M,), device=dev, dtype=torch.int32)
goff = torch.arange(G + 1, device=dev, dtype=torch.int32) * (M // G)
rs = torch.rand(M, device=dev)
out16 = torch.empty(M, D, device=dev, dtype=torch.float16)
w1 = torch.randn(G, F, D, device=dev) * D ** -0.5
w2 = torch.randn(G, D, F, device=dev) * F ** -0.5
b1 = torch.randn(G, F, device=dev) * 0.1
b2 = torch.randn(G, D, device=dev) * 0.1
pw1, pw2 = ops.PackedWeight(w1, fmt="f16"), ops.PackedWeight(w2, fmt="f16")


def run(shared, gathered):
    d = L.MlpDesc()
    d.h16 = L.H16_F16
    x = x16 if gathered else xbig
    d.X, d.ldx, d.gather = x.data_ptr(), D, (gather.data_ptr() if gathered else 0)
    d.M, d.Din, d.F, d.Dout = M, D, F, D
    d.goff, d.ngroups = goff.data_ptr(), G
    d.w1_gs, d.w2_gs = (0, 0) if shared else (F * pw1.Kp, D * pw2.Kp)
    d.b1_gs, d.b2_gs = F, D
    d.w1, d.ldw1, d.b1 = pw1.hi.data_ptr(), pw1.Kp, b1.data_ptr()
    d.w2, d.ldw2, d.b2 = pw2.hi.data_ptr(), pw2.Kp, b2.data_ptr()
    d.rowscale, d.r1_scale = rs.data_ptr(), 1.0
    d.C16, d.ldc = out16.data_ptr(), D
    return lambda: L.check(L.lib().mdm_fused_mlp(C.byref(d), C.c_void_p(L.stream_ptr())))


for rnd in range(2):
    for name, shared, gathered in (("16 distinct weight sets, gathered rows (the real case)", False, True),
                                   ("ONE shared weight set (L2-resident), gathered rows", True, True),
                                   ("16 distinct weight sets, contiguous rows", False, False),
                                   ("ONE shared weight set, contiguous rows", True, False)):
        print(f"round {rnd}: {name:58s}: {timeit(run(shared, gathered)):7.1f} us", flush=True)
