import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_bench import bench, L
for v, name in [(7, "glds"), (8, "regstage")]:
    L.lib().mdm_set_gemm_variant(v)
    for act in (0,):
        for (M, N, K) in [(128, 128, 512), (6272, 512, 512), (12544, 512, 512), (12544, 1536, 512), (12544, 2048, 512), (12544, 512, 2048), (50176, 1024, 512), (50176, 512, 1024), (8192, 8192, 8192)]:
            us, tf = bench(M, N, K, 1, act=act, a16=True, out16=True)
            print(f"{name:8s} act={act} M={M:6d} N={N:5d} K={K:5d}: {us:8.1f} us {tf:7.1f} TF")
