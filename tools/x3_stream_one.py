"""One expert-GEMM shape of the fp32-grade mode, 10 launches per kernel (for counter passes): the streamed-weight bf16x3 kernel
(csrc/gemm_stream3.hip) and the tile kernel (knob 69) on the same operands.  python tools/x3_stream_one.py M N K gelu(0|1)"""
import ctypes as C
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("motiondiffusion-moe_amd.ops")
L = importlib.import_module("motiondiffusion-moe_amd._lib")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from x3_stream_bench import x2_rows

M, N, K, act = (int(v) for v in sys.argv[1:5])
G, dev = 16, "cuda"
torch.manual_seed(0)
S = M // 4
x = x2_rows(torch.randn(S if act else M, K, device=dev))
w = torch.randn(G, N, K, device=dev) * K ** -0.5
b, rs = torch.randn(G, N, device=dev), torch.rand(M, device=dev)
pw, ws = ops.PackedWeight(w), ops.gemm_stream3x_pack(w)
goff = (torch.arange(G + 1, device=dev, dtype=torch.int64) * M // G).to(torch.int32)
gather = torch.randint(0, S, (M,), device=dev, dtype=torch.int32) if act else None
out, ox2 = torch.empty(M, N, device=dev), torch.empty(M, 2 * N, dtype=torch.bfloat16, device=dev)
for v in (0, 69):
    for _ in range(10):
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
torch.cuda.synchronize()
