"""In-kernel s_memtime stamps of one bf16 GEMM block: where does a K=512 block spend its time?"""
import importlib, sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ops = importlib.import_module("motiondiffusion-moe_amd.ops")
L = importlib.import_module("motiondiffusion-moe_amd._lib")
names = ["start", "ptrs+prefetch issued", "first stages issued", "tile0 landed", "loop done", "stores drained"]
for (M, N, K, v) in [(128, 128, 512, 1), (6272, 512, 512, 1), (12544, 512, 512, 1), (12544, 512, 512, 2), (12544, 2048, 512, 1), (12544, 512, 2048, 1)]:
    L.lib().mdm_set_gemm_variant(v)
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16); w = torch.randn(N, K, device="cuda") * K ** -0.5
    b = torch.randn(N, device="cuda"); r1 = torch.randn(M, N, device="cuda"); pw = ops.PackedWeight(w)
    out = torch.empty(M, N, device="cuda")
    d = ops.gemm_desc(1)
    d.A.p, d.A.ld, d.A.kind = x.data_ptr(), K, L.OP_BF16_ROW
    d.W = pw.operand(); d.M, d.N, d.K = M, N, K; d.C, d.ldc = out.data_ptr(), N; d.bias = b.data_ptr()
    d.R1, d.ldr1 = r1.data_ptr(), N
    d.feat_S = -77
    for _ in range(5): ops.run_gemm(d)
    torch.cuda.synchronize()
    st = (C.c_uint64 * 16)()
    L.check(L.lib().mdm_debug_stamps(st))
    t = [st[i] for i in range(9)]
    print(f"M={M} N={N} K={K} variant={v}: total {(t[5]-t[0])} cycles")
    for i in range(1, 5): print(f"    {names[i]:24s} +{t[i]-t[i-1]:8d}")
    print(f"    {'residuals issued + sync':24s} +{t[6]-t[4]:8d}")
    print(f"    {'staging writes + sync':24s} +{t[7]-t[6]:8d}")
    print(f"    {'LDS reads + stores issued':24s} +{t[8]-t[7]:8d}")
    print(f"    {'stores drained':24s} +{t[5]-t[8]:8d}")
