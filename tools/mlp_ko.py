"""Knock-out timings of the streamed-weight fused MLP (csrc/mlp_stream.hip): the kernel with one ingredient removed per
variant (knobs 41..46: wrong results, timing only), alternated with the real kernel in one process after a warm-up.
Runs on the DIAGNOSTIC library (python motiondiffusion-moe_amd/build.py --diag): the product library refuses those knobs."""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MDM_LIB", os.path.join(ROOT, "motiondiffusion-moe_amd", "libmdm_hip_diag.so"))
ops = importlib.import_module("motiondiffusion-moe_amd.ops")
L = importlib.import_module("motiondiffusion-moe_amd._lib")

NAMES = {0: "full", 34: "lds-staged kernel", 41: "no GELU arithmetic", 42: "no weight refills",           44: "no phase-1 MFMA", 45: "no phase-2 MFMA", 46: "no output stores", 47: "erf-form GELU", 48: "GELU interleaved with MFMAs"}


def timeit(fn, n=20):
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    dev, D, F, G = "cuda", 512, 1024, 16
    variants = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else list(NAMES)
    only16 = len(sys.argv) > 2 and sys.argv[2] == "only16"  # 16-bit output only, as the model's expert MLPs run
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
        # ragged groups like a real routing (+-10 %)
        sizes = torch.tensor([1.0 + 0.1 * ((i * 7) % 5 - 2) / 2 for i in range(G)])
        sizes = (sizes / sizes.sum() * M).long()
        sizes[-1] += M - sizes.sum()
        goff = torch.cat([torch.zeros(1, dtype=torch.long), sizes.cumsum(0)]).to(torch.int32).to(dev)
        rs = torch.rand(M, device=dev)

        def run(v):
            L.check(L.lib().mdm_set_gemm_variant(v))
            if only16:
                ops.fused_mlp(x16, pw1, b1, pw2, b2, gather=gather, goff=goff, rowscale=rs, rows=M, out16=out16, wstream=ws, only16=True)
            else:
                ops.fused_mlp(x16, pw1, b1, pw2, b2, gather=gather, goff=goff, rowscale=rs, rows=M, out=out, out16=out16, wstream=ws)
            L.lib().mdm_set_gemm_variant(0)

        for _ in range(100):
            run(0)
        res = {v: [] for v in variants}
        for _ in range(5):
            for v in variants:
                res[v].append(timeit(lambda: run(v)))
        fl = 4.0 * M * D * F
        for v in variants:
            t = sorted(res[v])[len(res[v]) // 2]
            print(f"M={M:6d} knob {v:2d} {NAMES.get(v, '?'):24s} {t:7.1f} us  ({fl / t / 1e6:6.0f} TF)  min {min(res[v]):7.1f}", flush=True)


if __name__ == "__main__":
    main()
