#!/usr/bin/env python3
"""Precision modes against the fp32-grade mode (bf16x3) on the bench workload, HIP vs HIP: error, routing decisions that
differ, optional kernel-selection knobs.   python tools/mode_compare.py [--variants 0,31,...] [--modes 4,2,1]"""
import argparse
import ctypes as C
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


def run(m, x, t, length, xf_proj, xf_out, L2):
    L = importlib.import_module("motiondiffusion-moe_amd._lib")
    B, T = x.shape[:2]
    dump = torch.full((L2, 2, B * T, 2), -1, dtype=torch.int32, device="cuda")
    L.lib().mdm_route_dump(C.c_void_p(dump.data_ptr()))
    y = m(x, t, length, xf_proj=xf_proj, xf_out=xf_out).clone()
    torch.cuda.synchronize()
    L.lib().mdm_route_dump(C.c_void_p(0))
    return y, dump.sort(-1).values


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="0")
    ap.add_argument("--modes", default="4,2,1")
    ap.add_argument("--config", default="small")
    ap.add_argument("--batch", type=int, default=32)
    a = ap.parse_args()
    L = importlib.import_module("motiondiffusion-moe_amd._lib")
    dev = torch.device("cuda")
    B, T, N = a.batch, 196, 28
    m3, inputs, _ = bench.build_model(a.config, dev, 3, B, T, N)
    x, length, xf_proj, xf_out = (v.to(dev) for v in inputs)
    t = torch.full((B,), 977, dtype=torch.int64, device=dev)
    L2 = 2 * m3.num_layers
    ref, rref = run(m3, x, t, length, xf_proj, xf_out, L2)
    valid = rref[..., 0] >= 0
    del m3
    for prec in [int(v) for v in a.modes.split(",")]:
        m, _, _ = bench.build_model(a.config, dev, prec, B, T, N)
        for var in [int(v) for v in a.variants.split(",")]:
            L.lib().mdm_set_gemm_variant(var)
            y, r = run(m, x, t, length, xf_proj, xf_out, L2)
            L.lib().mdm_set_gemm_variant(0)
            d = (y - ref).abs()
            frame = d.amax(-1) / ref.abs().amax()
            flips = int(((r != rref).any(-1) & valid).sum())
            print(f"precision {prec} variant {var}: rel err {float(d.max() / ref.abs().max()):.2e}  median frame "
                  f"{float(frame.median()):.2e}  frames > 5 %: {int((frame > 0.05).sum())}/{frame.numel()}  "
                  f"routing decisions that differ: {flips}/{int(valid.sum())}", flush=True)
        del m
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
