"""A/B of the GEMM tile variants on the big model's shapes (16-bit operands, fp32 + 16-bit outputs, GELU epilogue,
graph-timed): knob 7 = 128^2 tiles only, 6 = the 256^2 tile wherever eligible."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
from gemm_bench import bench, L

shapes = [(12544, 1024, 1024), (6272, 1024, 1024), (12544, 3072, 1024), (12544, 4096, 1024), (12544, 1024, 4096),
          (50176, 2048, 1024), (50176, 1024, 2048), (12544, 512, 512), (12544, 2048, 512), (8192, 8192, 8192)]
for v, name in [(7, "128^2 tiles"), (6, "256^2, BK64 x2")]:
    L.lib().mdm_set_gemm_variant(v)
    print(f"variant {v} ({name}):")
    for (M, N, K) in shapes:
        us, tf = bench(M, N, K, 1, act=1, a16=True, out16=True)
        print(f"  M={M:6d} N={N:5d} K={K:5d}: {us:8.1f} us {tf:7.1f} TF", flush=True)
L.lib().mdm_set_gemm_variant(0)
