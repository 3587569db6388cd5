"""Time the MoE-block training step (forward + backward + clip + Adam) at the bench shape: B=64 rows of T=196 (M = 12544
tokens), D=512, F=1024, E=8.  Usage: python tools/moe_train_bench.py [B S D F E]"""
import importlib
import sys
import time

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mt = importlib.import_module("motiondiffusion-moe_amd.moe_train")
B, S, D, F, E = [int(a) for a in sys.argv[1:6]] if len(sys.argv) >= 6 else (64, 196, 512, 1024, 8)
Te = 4 * D
tr = mt.MoEFFNTrainer(D, F, E, Te)
g = torch.Generator(device="cuda").manual_seed(0)
tr.params.flat.copy_((torch.rand(tr.params.flat.numel(), device="cuda", generator=g) * 2 - 1) * 0.05)
tr.params.views["ln_w"].add_(1.0), tr.params.views["st_norm_w"].add_(1.0)
x = torch.randn(B, S, D, device="cuda", generator=g)
emb = torch.randn(B, D, device="cuda", generator=g)
eph = (torch.randn(Te, D, device="cuda", generator=g) * D ** -0.5, torch.zeros(Te, device="cuda"))
tgt = torch.randn(B, S, D, device="cuda", generator=g)
for phase in ("warm", "timed"):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 3 if phase == "warm" else 10
    tf = tb = to = 0.0
    for _ in range(n):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        e[0].record()
        out = tr.forward(x, emb, eph)
        e[1].record()
        tr.backward(out - tgt)
        e[2].record()
        tr.optimizer_step()
        e[3].record()
        torch.cuda.synchronize()
        tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2]); to += e[2].elapsed_time(e[3])
    if phase == "timed":
        M = B * S
        flop_f = 2 * 4 * M * D * F * 2 + 2 * M * D * D
        flop = 3 * flop_f  # forward + data gradients + weight gradients
        tot = (tf + tb + to) / n
        print(f"MoE block training step B*S={M} D={D} F={F} E={E}: forward {tf / n:.2f} ms, backward {tb / n:.2f} ms, clip+Adam {to / n:.3f} ms"
              f" -> {tot:.2f} ms/step, {flop / tot / 1e9:.1f} TFLOP/s algorithmic (bf16x3 arithmetic = 3 MFMA passes per product)")
