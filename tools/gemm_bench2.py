import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_bench import bench, L
for (M, N, K) in [(6272, 512, 512), (12544, 512, 512), (12544, 1536, 512), (12544, 2048, 512), (12544, 512, 2048), (6272, 1536, 512), (6272, 2048, 512), (6272, 512, 2048)]:
    line = f"M={M:6d} N={N:5d} K={K:5d}:"
    for v, name in [(1, "BM128"), (2, "BM64")]:
        L.lib().mdm_set_gemm_variant(v)
        for act in (0, 1):
            us, tf = bench(M, N, K, 1, act=act, a16=True, out16=False)
            line += f"  {name} act{act} {us:6.1f} us"
    print(line, flush=True)
