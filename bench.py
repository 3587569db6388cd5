#!/usr/bin/env python3
"""Headline benchmark: denoising-steps/sec of the CFG DDPM sampling step (BASELINE.json configs[1]:
model_size=small, 8 experts, B=32 per GPU, T=196, 263-d, 1000-step schedule, bf16 MFMA).

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One step = the whole hot path over one batch: cond+uncond rows (2B) through the denoiser, guidance on pred_xstart,
posterior update, noise.  Inputs are resident in HBM before the timed region.  Prints ONE JSON line on rank 0.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

CONFIGS = {
    # name: (ctor kwargs, algorithmic FLOP per forward at B=32,T=196,N=28 from SURVEY.md §8d, MoE share)
    "small": (dict(latent_dim=512, ff_size=1024, num_layers=4, num_heads=4, text_latent_dim=256, moe_num_experts=8,
                   model_size="small"), 0.9012e12, 0.338e12),
    "big": (dict(latent_dim=512, ff_size=1024, num_layers=4, num_heads=4, text_latent_dim=256, moe_num_experts=8,
                 model_size="big"), 3.5926e12, 1.349e12),
    # configs[4]: big, 16 experts, top-2, fp8 expert GEMMs (run with --precision 5 --batch 8: B=64 over 8 GPUs)
    "big16": (dict(latent_dim=512, ff_size=1024, num_layers=4, num_heads=4, text_latent_dim=256, moe_num_experts=16,
                   model_size="big"), 3.5938e12, 1.349e12),
}
# dense bf16 / fp16 MFMA peak (same rate); the bf16x3 mode issues 3 MFMAs per product; the mixed mode issues 3 per product
# except in the expert MLPs + 4x FFN (52 % of the FLOPs at the small config: SURVEY.md section 8a): blended 1 / (0.48*3 + 0.52)
# precision 5 (fp8 expert GEMMs): blended by FLOP share as SURVEY.md section 8(d) prescribes: the MoE FFN's 37.5 % at the 5 PF
# dense fp8 rate, the rest at 2.5 PF -> 1 / (0.375 / 5 + 0.625 / 2.5) = 3.08 PF
PEAK = {1: 2.5e15, 2: 2.5e15, 3: 2.5e15 / 3, 4: 2.5e15 / (0.48 * 3 + 0.52), 5: 1.0 / (0.375 / 5.0e15 + 0.625 / 2.5e15)}
DTYPE = {1: "bf16", 2: "f16", 3: "bf16x3(fp32-grade)", 4: "mixed(bf16x3 + f16 expert/FFN GEMMs)", 5: "f16 + fp8(e4m3) expert GEMMs"}


_SD_CACHE = {}  # (config, seed) -> synthetic state dict of the LAST config built (the big models' take 15 - 25 s of host time each)


def build_model(cfg_name, device, precision, B, T, N, seed=0, N_u=None):
    T_ = importlib.import_module("motiondiffusion-moe_amd.transformer")
    synth = importlib.import_module("motiondiffusion-moe_amd.synth")
    kw, _, _ = CONFIGS[cfg_name]
    # every parameter is overwritten by the synthetic state dict below: skip the constructor's random init (15 s for the big model)
    reset = T_.MotionTransformer.reset_parameters
    T_.MotionTransformer.reset_parameters = lambda self: None
    try:
        m = T_.MotionTransformer(263, num_frames=196, precision=precision, **kw)
    finally:
        T_.MotionTransformer.reset_parameters = reset
    if (cfg_name, seed) not in _SD_CACHE:
        _SD_CACHE.clear()
        _SD_CACHE[(cfg_name, seed)] = synth.synth_state_dict(m._layout, seed)
    sd = _SD_CACHE[(cfg_name, seed)]
    m.load_state_dict(sd, strict=True)
    D, Dt, L = m.latent_dim, m.text_latent_dim, m.num_layers
    eph = synth.synth_ephemerals(D, Dt, L, 7)
    proj = synth.synth_projections(D // m.num_heads, L, 7)
    m.set_ephemerals(eph), m.set_projections(proj)
    x, _, length, xf_proj, xf_out = synth.synth_inputs(B, T, 263, N, Dt, seed, min_len=40)
    # the empty caption's embedding; with its own token count (N_u) the sampler pads it and passes per-row token counts
    xo_u = synth.uniform_pm1((1, N_u or N, Dt), "in.uncond", seed) * (3.0 ** 0.5)
    m = m.to(device).eval()
    m.set_uncond_embedding(xo_u.mean(1).to(device), xo_u.to(device))
    host = dict(sd=sd, eph={n: (w, b) for n, w, b in eph}, proj=dict(proj), xo_u=xo_u,
                mcfg=dict(latent_dim=D, num_heads=m.num_heads, num_layers=L, moe_num_experts=m.moe_num_experts))
    return m, (x, length, xf_proj, xf_out), host


def usable_cores() -> int:
    """Threads the CPU leg may really use: the cgroup CPU quota when there is one (a GPU box exposes every host core in
    the affinity mask but schedules only its share), else the affinity mask, never more than 16 (the documented
    per-GPU CPU share of the pool); MDM_CPU_THREADS overrides."""
    if os.environ.get("MDM_CPU_THREADS"):
        return max(1, int(os.environ["MDM_CPU_THREADS"]))
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except (OSError, ValueError):
            pass
    return max(1, min(n, 16))


def cpu_baseline(host, inputs, steps_total, cfg_scale, sample_rows=16, nsteps=2):
    """The oracle (CPU restatement, pinned to the reference by golden vectors) timed on this host: one CFG step
    (2 forwards) on a bounded sample of the batch, scaled linearly to the full batch (samples are independent)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import denoiser_ref as R
    import diffusion_ref as DR
    x, length, xf_proj, xf_out = inputs
    B = x.shape[0]
    n = min(sample_rows, B)
    cores = usable_cores()
    torch.set_num_threads(cores)
    print(f"[bench] cpu_baseline: timing the oracle on {n}/{B} samples with {cores} threads ...", file=sys.stderr, flush=True)
    xo_u = host["xo_u"].expand(n, -1, -1).contiguous()
    xp_u = xo_u.mean(1)
    tb = DR.Tables(DR.linear_betas(steps_total))
    t = steps_total - 1
    tt = torch.full((n,), t, dtype=torch.int64)
    xs, ls = x[:n], length[:n]

    def fwd(cond, rows=None):
        rows = n if rows is None else rows
        xp, xo = (xf_proj[:n], xf_out[:n]) if cond else (xp_u, xo_u)
        return R.denoiser_forward(host["sd"], host["mcfg"], xs[:rows], tt[:rows], ls[:rows], xp[:rows], xo[:rows],
                                  host["eph"], host["proj"])

    with torch.no_grad():
        fwd(True, rows=1)  # page in weights / warm the thread pool
        t0 = time.perf_counter()
        for i in range(nsteps):
            xs, _ = DR.cfg_step(tb, t, xs, fwd(True), fwd(False), torch.zeros_like(xs), cfg_scale)
            print(f"[bench] cpu_baseline: step {i + 1}/{nsteps} done at {time.perf_counter() - t0:.1f} s", file=sys.stderr, flush=True)
        dt = (time.perf_counter() - t0) / nsteps
    full = dt * B / n
    print(f"[bench] cpu_baseline: {dt:.1f} s for the sample -> {full:.1f} s per full step", file=sys.stderr, flush=True)
    try:
        cpu_model = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
    except (OSError, StopIteration):
        cpu_model = "unknown"
    return {"value": 1.0 / full, "unit": "denoising-steps/sec", "cores": cores, "kind": "port", "cpu_model": cpu_model,
            "sample": f"{nsteps} CFG steps (2 forwards each) of the torch-CPU oracle (fp32) on {n} of the {B} samples, "
                      f"T={x.shape[1]}, {dt:.1f} s per step measured" + (f", scaled x{B / n:g} to the batch" if n != B else "")}


def source_sha16() -> str:
    """sha256 over the kernel sources (csrc/*.hip, csrc/*.h, include/*.h): ties a committed counter profile to the code it was
    measured on (tools/pmc_traffic.py stores the same hash)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    pk = os.path.join(ROOT, "motiondiffusion-moe_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(pk, "*.hip")) + glob.glob(os.path.join(pk, "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def committed_traffic(precision: int):
    """Fabric bytes from the committed rocprofv3 --pmc passes (profiles/r04_pmc_traffic_p<precision>.json), or (None, why).  The
    profile is used only if it was taken on exactly these kernel sources; a stale one is reported as stale, not printed."""
    f = os.path.join(ROOT, "profiles", f"r04_pmc_traffic_p{precision}.json")
    if not os.path.exists(f):
        return None, f"no counter profile committed for precision {precision}"
    j = json.load(open(f))
    if j.get("src_sha16") != source_sha16():
        return None, (f"profiles/{os.path.basename(f)} was measured at commit {j.get('commit', '?')} on kernel sources "
                      f"{j.get('src_sha16')}, the sources are now {source_sha16()}: stale, not reported")
    return j, (f"fabric bytes from rocprofv3 --pmc FETCH_SIZE(x2) / WRITE_SIZE passes, profiles/{os.path.basename(f)}, measured at commit "
               f"{j.get('commit', '?')} on these kernel sources ({j.get('src_sha16')})")


def time_block(fn, iters=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


def moe_block_rate(m, B2, T, precision):
    """HIP-event timing of the MoE FFN block (router + grouped expert GEMMs + combine/stylization) alone, on the
    stream it is launched on; algorithmic FLOP = 16*M*D*F + 4*M*D*E + style at M = B2*T rows."""
    import ctypes as C
    L = importlib.import_module("motiondiffusion-moe_amd._lib")
    pm = m.pack()
    D, F_, E = m.latent_dim, m.ff_size, m.moe_num_experts
    dev = m.device
    M = B2 * T
    h = torch.randn(B2, T, D, device=dev)
    sc = torch.randn(4, B2, 2 * D, device=dev) * 0.1
    ln = torch.full((B2,), T, dtype=torch.int32, device=dev)
    out = torch.empty_like(h)
    ws = m._workspace(B2, T, 28)
    layer = m.num_layers  # first full-scale layer

    def run():
        L.check(L.lib().mdm_block_forward(C.byref(pm.model), C.c_int32(layer), C.c_int32(L.BLOCK_MOE), C.c_void_p(0),
                                          C.c_void_p(h.data_ptr()), C.c_void_p(sc.data_ptr()), C.c_void_p(ln.data_ptr()),
                                          C.c_int32(B2), C.c_int32(T), C.c_void_p(out.data_ptr()), C.c_void_p(ws.data_ptr()),
                                          C.c_int64(ws.numel()), C.c_void_p(0), C.c_int32(precision), C.c_void_p(L.stream_ptr())))

    saved = {k: v.clone() for k, v in m.moe_buffers().items()}
    dt = time_block(run)
    for k, v in saved.items():
        m.moe_buffers()[k].copy_(v)
    flop = 16.0 * M * D * F_ + 4.0 * M * D * E + 2.0 * M * D * D
    return dt, flop


def expert_mlp_rate(m, B2, T):
    """HIP-event timing of the dominant kernel alone (fused expert MLP, csrc/mlp.hip) at the step's shape: 4*M routed
    rows (2 branches x top-2) spread evenly over 2*E expert slabs, gathered from 2*M normalised rows.  Algorithmic FLOP
    per launch = 4 * rows * D * F (two GEMMs).  Returns (seconds, flop) or None when the shape is not covered."""
    ops = importlib.import_module("motiondiffusion-moe_amd.ops")
    D, F_, E = m.latent_dim, m.ff_size, m.moe_num_experts
    if D != 512 or F_ % 256 or m.precision not in (1, 2):
        return None
    h16 = torch.float16 if m.precision == 2 else torch.bfloat16
    fmt = "f16" if m.precision == 2 else "bf16"
    dev, M = m.device, B2 * T
    rows, G = 4 * M, 2 * E
    g = torch.Generator(device="cpu").manual_seed(0)
    x16 = torch.randn(2 * M, D, generator=g).to(dev).to(h16)
    w1s, w2s = (torch.randn(G, F_, D, generator=g) * D ** -0.5).to(dev), (torch.randn(G, D, F_, generator=g) * F_ ** -0.5).to(dev)
    w1, w2 = ops.PackedWeight(w1s, fmt=fmt), ops.PackedWeight(w2s, fmt=fmt)
    b1, b2 = torch.zeros(G, F_, device=dev), torch.zeros(G, D, device=dev)
    gather = torch.randint(0, 2 * M, (rows,), generator=g, dtype=torch.int32).to(dev)
    goff = (torch.arange(G + 1, dtype=torch.int64) * rows // G).to(torch.int32).to(dev)
    rs = torch.rand(rows, generator=g).to(dev)
    out16 = torch.empty(rows, D, device=dev, dtype=h16)  # 16-bit output only: what the model's expert MLPs write in this mode
    ws = ops.mlp_stream_pack(w1s, w2s, h16)  # the weight stream the model's packer builds (csrc/mlp_stream.hip)
    dt = time_block(lambda: ops.fused_mlp(x16, w1, b1, w2, b2, gather=gather, goff=goff, rowscale=rs, rows=rows, out16=out16,
                                          wstream=ws, only16=True))
    return dt, 4.0 * rows * D * F_


def probe_dominant_kernel(r, m, steps=3):
    """Live per-launch durations of the dominant kernel inside REAL sampling steps: HIP events recorded by the library
    on the launch stream around each fused expert-MLP launch (include/mdm_hip.h: mdm_probe_*), during eager (uncaptured)
    steps run after the timed region on the same state.  Returns (mean seconds per launch, mean algorithmic FLOP per
    launch, list of (rows, us)) or None when the kernel is not on this configuration's path."""
    import ctypes as C
    L = importlib.import_module("motiondiffusion-moe_amd._lib")
    lib = L.lib()
    D, F_ = m.latent_dim, m.ff_size
    r._step(True)  # warm the eager path
    L.check(lib.mdm_probe_enable(1))
    for _ in range(steps):
        r._step(True)
    torch.cuda.synchronize()
    us = (C.c_float * 64)()
    rows = (C.c_int32 * 64)()
    n = lib.mdm_probe_read(us, rows, 64)
    L.check(lib.mdm_probe_enable(0))
    if n <= 0:
        return None
    n = min(n, 64)
    pairs = [(int(rows[i]), float(us[i])) for i in range(n)]
    t = sum(u for _, u in pairs) * 1e-6 / n
    flop = sum(4.0 * rw * D * F_ for rw, _ in pairs) / n  # two GEMMs: 2 * rows * D * F each
    return t, flop, pairs


def mode_table(a, m_main, inputs, host, diff, kw, dev, main_prec, main_ms, steps=6, warmup=2):
    """Every precision mode on the SAME workload: ms per guided step (graph replay) and, for one forward of the cond half,
    the error and the routing decisions that differ relative to the parity-grade mode (bf16x3; its own error against the
    CPU oracle is gated in tests/test_round2_gpu.py).  No oracle is involved here: HIP against HIP on identical inputs."""
    import ctypes as C
    L = importlib.import_module("motiondiffusion-moe_amd._lib")
    x, length, xf_proj, xf_out = inputs
    B, T = x.shape[0], x.shape[1]
    L2 = 2 * m_main.num_layers
    t = torch.full((B,), a.schedule - 23, dtype=torch.int64, device=dev)
    xd, ld = x.to(dev), length.to(dev)
    res, outs, routes = {}, {}, {}
    for prec in (3, 4, 2, 1):
        if prec == main_prec:
            m = m_main
        else:
            m, _, _ = build_model(a.config, dev, prec, B, T, xf_out.shape[1], seed=0)
        dump = torch.full((L2, 2, B * T, 2), -1, dtype=torch.int32, device=dev)
        L.lib().mdm_route_dump(C.c_void_p(dump.data_ptr()), C.c_int64(dump.numel()))
        try:  # the dump pointer is process-global in the library: never leave it pointing at a freed tensor
            outs[prec] = m(xd, t, ld, xf_proj=kw["xf_proj"], xf_out=kw["xf_out"]).clone()
            torch.cuda.synchronize()
        finally:
            L.lib().mdm_route_dump(C.c_void_p(0), C.c_int64(0))
        routes[prec] = dump.sort(-1).values
        if prec == main_prec:
            ms = main_ms
        else:
            r = diff._runner(m, (B, T, 263), kw, dev, "cfg", a.cfg_scale, 0.0, False, not a.no_graph, 1)
            r.philox = (1234, 0)
            r._prepare()
            if r.use_graph:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    r._step(True)
                r.graph = g
            r.xx[:B].copy_(r.draw_xT(1234, 0))
            r.t_dev.fill_(a.schedule - 1)
            step = (lambda: r.graph.replay()) if r.graph is not None else (lambda: r._step(True))
            for _ in range(warmup):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / steps * 1e3
            live3 = None
            if prec == 3:  # the parity mode's own roofline entry: its dominant kernels, timed live inside real (eager) steps
                r.t_dev.fill_(a.schedule - 1)
                live3 = probe_dominant_kernel(r, m, steps=2)
            del r
        res[prec] = {"precision": prec, "dtype": DTYPE[prec], "ms_per_step": round(ms, 3), "steps_per_s": round(1e3 / ms, 2)}
        if prec == 3 and prec != main_prec:
            flop_step = 2.0 * CONFIGS[a.config][1] * (B / 32.0) * (T / 196.0)
            res[3]["roofline"] = {
                "bound": "mfma", "peak": round(PEAK[3] / 1e12, 1), "unit": "TFLOP/s",
                "peak_note": "2.5 PF dense bf16 / 3: every fp32-grade product is three bf16 MFMAs (hi*lo + lo*hi + hi*hi)",
                "whole_step": {"achieved": round(flop_step / (ms * 1e-3) / 1e12, 2), "frac": round(flop_step / (ms * 1e-3) / PEAK[3], 4)}}
            if live3 is not None:
                res[3]["roofline"].update({
                    "achieved": round(live3[1] / live3[0] / 1e12, 2), "frac": round(live3[1] / live3[0] / PEAK[3], 4),
                    "kernel": "gemm_stream3_kernel<7, 512 | 1024, GELU | none> x 2: the expert MLP as its two grouped, streamed-weight bf16x3 "
                              "GEMMs (csrc/gemm_stream3.hip), the dominant kernels of this mode",
                    "what": "mean algorithmic FLOP per expert-MLP launch pair (4 * routed rows * D * F) / mean duration of the pair, HIP "
                            "events on the launch stream inside real sampling steps (mdm_probe_*)",
                    "launch_us_mean": round(live3[0] * 1e6, 1), "launches_timed": len(live3[2]),
                    "launches": [{"rows": rw, "us": round(u, 1)} for rw, u in live3[2][:8]]})
        if prec != main_prec:
            del m
            torch.cuda.empty_cache()
    ref, rref = outs[3], routes[3]
    valid = int((rref[..., 0] >= 0).sum())
    for prec in (4, 2, 1):
        d = (outs[prec] - ref).abs()
        frame = d.amax(-1) / ref.abs().amax()
        res[prec].update({
            "rel_err_vs_parity_mode": float(d.max() / ref.abs().max()),
            "median_frame_err_vs_parity_mode": float(frame.median()),
            "frames_off_by_5pct": int((frame > 0.05).sum()), "frames": int(frame.numel()),
            "routing_decisions_that_differ": int(((routes[prec] != rref).any(-1) & (rref[..., 0] >= 0)).sum()),
            "routing_decisions": valid})
    res[3]["rel_err_vs_cpu_oracle"] = "<= 1e-3 gated (7e-5 measured) in tests/test_round2_gpu.py::test_configs1_free_routing_error_and_flip_budget"
    out = {DTYPE[p].split("(")[0]: v for p, v in res.items() if p != 3}
    out["parity_mode"] = res[3]
    return out


def spawn_ranks(n: int) -> int:
    """Start `python -m torch.distributed.run --nnodes=1 --nproc-per-node n ... bench.py <same arguments>` as a child process,
    relay its output, return its exit code (non-zero when any rank failed).  The caller has not initialised a GPU."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this pool
    print(f"[bench] launching {n} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in proc.stdout:
        sys.stdout.write(line)
        sys.stdout.flush()
    return proc.wait()


def dry_run(a, world, rank) -> None:
    """The bench protocol without the GPU (tests/test_bench_launcher.py): gloo process group, W warm-up + K timed stand-in
    steps between barriers, MAX over ranks, one all-gather of every rank's shard, ONE JSON line from rank 0."""
    if world > 1:
        dist.init_process_group("gloo")
    dmod = importlib.import_module("motiondiffusion-moe_amd.dist")
    B = a.batch
    lo, hi = dmod.shard_range(B * world, rank, world)
    x = torch.arange(lo, hi, dtype=torch.float32).reshape(-1, 1, 1).expand(-1, 4, 3).contiguous()

    def one_step():
        x.mul_(1.0)

    for _ in range(a.warmup):
        one_step()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        one_step()
    if world > 1:
        dist.barrier()
    tmax = torch.tensor([time.perf_counter() - t0])
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    final = dmod.all_gather_ragged(x, B * world)
    assert final.shape[0] == B * world and torch.equal(final[:, 0, 0], torch.arange(B * world, dtype=torch.float32))
    if rank == 0:
        print(json.dumps({"metric": "denoising-steps/sec (B=32, T=196, 263-d, 8 experts)", "value": a.steps / dt * world * (B / 32.0),
                          "unit": "denoising-steps/sec", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
                          "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                          "dtype": "none", "data": "synthetic", "dry_run": True,
                          "config": {"workload": "dry run: stand-in step on the CPU, gloo ranks (launcher rehearsal)",
                                     "global_batch": B * world, "parallelism": f"batch-shard x{world}, weights replicated"}}),
              flush=True)
    if world > 1:
        dist.destroy_process_group()


def other_config_line(cfg_name, sampler, schedule, a, dev, steps=8, warmup=2, batch=None, precision=None):
    """One more BASELINE config timed in the same process (graph-replayed steps on the same device; the headline's precision and
    per-GPU batch unless given): the driver's record then carries the big model too.  Returns the numbers, not a full bench line."""
    D_ = importlib.import_module("motiondiffusion-moe_amd.diffusion")
    B, T, N = batch or a.batch, a.frames, a.text_tokens
    prec = precision or a.precision
    m, inputs, _ = build_model(cfg_name, dev, prec, B, T, N, seed=0)
    x, length, xf_proj, xf_out = inputs
    diff = D_.GaussianDiffusion(betas=D_.get_named_beta_schedule("linear", schedule), model_mean_type=D_.ModelMeanType.EPSILON,
                                model_var_type=D_.ModelVarType.FIXED_SMALL, loss_type=D_.LossType.MSE)
    kw = {"xf_proj": xf_proj.to(dev), "xf_out": xf_out.to(dev), "length": length.to(dev), "text": ["synthetic"] * B}
    r = diff._runner(m, (B, T, 263), kw, dev, sampler, a.cfg_scale, 0.0, False, True, 1)
    r.philox = (1234, 0)
    r._prepare()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        r._step(True)
    r.xx[:B].copy_(r.draw_xT(1234, 0))
    r.t_dev.fill_(schedule - 1)
    for _ in range(warmup):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        g.replay()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    assert bool(torch.isfinite(r.xx[:B]).all())
    _, flop_fwd, _ = CONFIGS[cfg_name]
    nfwd = 2.0 if sampler == "cfg" else 1.0
    flop_step = nfwd * flop_fwd * (B / 32.0) * (T / 196.0)
    out = {"workload": f"model_size=big, num_experts={m.moe_num_experts}, B={B}, T={T}, "
                       + (f"{schedule}-step DDPM with CFG {a.cfg_scale} (2B rows per forward)" if sampler == "cfg"
                          else f"{schedule}-step DDIM (one forward per step)") + ", hipGraph=on",
           "ms_per_step": round(ms, 3), "steps_per_s": round(1e3 / ms, 2), "dtype": DTYPE[prec], "steps": steps, "warmup": warmup,
           "whole_step": {"achieved": round(flop_step / (ms * 1e-3) / 1e12, 2), "unit": "TFLOP/s",
                          "frac": round(flop_step / (ms * 1e-3) / PEAK[prec], 4)}}
    del r, g, m
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="small", choices=list(CONFIGS))
    ap.add_argument("--precision", type=int, default=1, choices=[1, 2, 3, 4, 5],
                    help="1 = bf16 MFMA (default: the format BASELINE configs[1] names), 2 = fp16 MFMA (same speed, 8x smaller error), "
                         "3 = bf16x3 fp32-grade (the only mode inside the 1e-3 parity bound: always reported beside the headline as "
                         "parity_mode), 4 = mixed (bf16x3 + fp16 expert/FFN GEMMs)")
    ap.add_argument("--no-modes", action="store_true", help="skip the per-mode timing / error table")
    ap.add_argument("--sampler", default="cfg", choices=["cfg", "ddim"],
                    help="cfg = guided DDPM step (2 forwards batched; configs[1], [2]); ddim = DDIM step (1 forward; configs[3])")
    ap.add_argument("--batch", type=int, default=32, help="samples per GPU")
    ap.add_argument("--frames", type=int, default=196)
    ap.add_argument("--text-tokens", type=int, default=28, help="text tokens per caption (the reference pads to 8 + 77 = 85)")
    ap.add_argument("--uncond-tokens", type=int, default=0,
                    help="text tokens of the empty caption when they differ from --text-tokens (a real tokenizer gives it 8 + 2); "
                         "0 = the same count")
    ap.add_argument("--schedule", type=int, default=1000)
    ap.add_argument("--cfg-scale", type=float, default=7.5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--streams", type=int, default=1, help="concurrent HIP streams the step's rows are split over")
    ap.add_argument("--variant", type=int, default=0, help="kernel-selection knob for same-box A/B runs (mdm_set_gemm_variant)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the configs[2] / configs[3] timings (big model) of the N = 1 line")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / timing-protocol rehearsal on the CPU: gloo ranks, a stand-in step, no GPU and no model")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: this process only LAUNCHES (it never touches a GPU); the ranks are N fresh
        # processes under torch.distributed.run, one per GPU, and rank 0's JSON line comes back through our stdout
        raise SystemExit(spawn_ranks(a.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(a.gpus, 1):
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {a.gpus}")
    if a.dry_run:
        return dry_run(a, world, rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the denoising path runs on hand-written HIP kernels only")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)
    D_ = importlib.import_module("motiondiffusion-moe_amd.diffusion")
    dmod = importlib.import_module("motiondiffusion-moe_amd.dist")

    if a.variant:
        importlib.import_module("motiondiffusion-moe_amd._lib").lib().mdm_set_gemm_variant(a.variant)
    B, T, N = a.batch, a.frames, a.text_tokens
    m, inputs, host = build_model(a.config, dev, a.precision, B, T, N, seed=0, N_u=a.uncond_tokens or None)
    x, length, xf_proj, xf_out = inputs
    diff = D_.GaussianDiffusion(betas=D_.get_named_beta_schedule("linear", a.schedule),
                                model_mean_type=D_.ModelMeanType.EPSILON, model_var_type=D_.ModelVarType.FIXED_SMALL,
                                loss_type=D_.LossType.MSE)
    kw = {"xf_proj": xf_proj.to(dev), "xf_out": xf_out.to(dev), "length": length.to(dev), "text": ["synthetic"] * B}
    r = diff._runner(m, (B, T, 263), kw, dev, a.sampler, a.cfg_scale, 0.0, False, not a.no_graph, a.streams)
    r.philox = (1234, dmod.shard_range(B * world, rank, world)[0])
    r._prepare()
    if r.use_graph:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            r._step(True)
        r.graph = g
    # world-size independent noise: x_T and every step's noise are functions of (seed, GLOBAL sample index, timestep,
    # element) from the counter-based device generator (csrc/noise.hip); the step's noise draw is part of the captured graph
    lo, hi = dmod.shard_range(B * world, rank, world)
    r.xx[:B].copy_(r.draw_xT(1234, lo))
    r.t_dev.fill_(a.schedule - 1)
    done = [0]

    def one_step():
        if done[0] and done[0] % a.schedule == 0:
            r.t_dev.fill_(a.schedule - 1)  # a run longer than the schedule starts over instead of underflowing t
        done[0] += 1
        if r.graph is not None:
            r.graph.replay()
        else:
            r._step(True)

    for _ in range(a.warmup):
        one_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        one_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    # the path's single collective: gather the motion tensor of every shard (RCCL over xGMI)
    final = dmod.all_gather_ragged(r.xx[:B].clone(), B * world)
    assert final.shape[0] == B * world and bool(torch.isfinite(final).all())

    if rank == 0:
        _, flop_fwd, flop_moe = CONFIGS[a.config]
        scale = (B / 32.0) * (T / 196.0)
        nfwd = 2.0 if a.sampler == "cfg" else 1.0
        flop_step = nfwd * flop_fwd * scale  # cond + uncond forwards (CFG) or one forward (DDIM)
        ms = dt / a.steps * 1e3
        steps_per_s = a.steps / dt
        value = steps_per_s * world * (B / 32.0)
        achieved = flop_step / (dt / a.steps)
        rows_b = (2 if a.sampler == "cfg" else 1) * B
        moe_dt, moe_flop = moe_block_rate(m, rows_b, T, a.precision)
        dom = expert_mlp_rate(m, rows_b, T)
        done[0] = 0
        r.t_dev.fill_(a.schedule - 1)
        live = probe_dominant_kernel(r, m) if not r.chunks else None  # single-stream steps only
        # (the per-mode table is a single-GPU report: a scaling run keeps rank 0 no longer than the other ranks)
        modes = None if (a.no_modes or world > 1 or a.sampler != "cfg" or a.config != "small") else mode_table(a, m, inputs, host, diff, kw, dev, a.precision, ms)
        # fabric bytes per step from the committed rocprofv3 --pmc passes -- only for this exact workload AND these kernel sources
        pmc_j, traffic_note = (None, "the committed counter profiles are of the default workload (small, B=32, T=196, CFG)")
        if (a.config, B, T, a.sampler) == ("small", 32, 196, "cfg"):
            pmc_j, traffic_note = committed_traffic(a.precision)
        traffic = pmc_j["total_bytes_per_step"] if pmc_j else None
        cfgname = {"small": "configs[1]", "big": "configs[2]" if a.sampler == "cfg" else "configs[3]", "big16": "configs[4]"}[a.config]
        stepdesc = (f"{a.schedule}-step DDPM with CFG {a.cfg_scale} (cond+uncond batched as {2 * B} rows)" if a.sampler == "cfg"
                    else f"{a.schedule}-step DDIM (eta 0, one forward per step)")
        workload = (f"{cfgname}: model_size={'small' if a.config == 'small' else 'big'}, num_experts={m.moe_num_experts}, B={B}/GPU, T={T}, {stepdesc}, N_text={N}, "
                    f"hipGraph={'on' if r.graph is not None else 'off'}, streams={r.nstreams if r.chunks else 1}")
        line = {
            "metric": "denoising-steps/sec (B=32, T=196, 263-d, 8 experts)", "value": round(value, 3),
            "unit": "denoising-steps/sec", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": DTYPE[a.precision], "data": "synthetic",
            "config": {"workload": workload,
                       "global_batch": B * world, "parallelism": f"batch-shard x{world}, weights replicated"},
            "sample_steps_per_s": round(steps_per_s * B * world, 1),
            # SURVEY 8(d) asks for ms per model forward as well: a CFG step is two B-row forwards (run here as ONE 2B-row launch
            # sequence) plus the guidance / posterior update, a DDIM step is one
            "ms_per_forward": round(ms / nfwd, 3),
            "roofline": {"bound": "mfma", "achieved": round(achieved / 1e12, 2), "peak": PEAK[a.precision] / 1e12,
                         "unit": "TFLOP/s", "frac": round(achieved / PEAK[a.precision], 4), "traffic": traffic,
                         "traffic_note": traffic_note,
                         "what": "whole step: algorithmic FLOP of 2 forwards / wall time per step",
                         "whole_step": {"achieved": round(achieved / 1e12, 2), "frac": round(achieved / PEAK[a.precision], 4),
                                        "traffic": traffic},
                         "moe_ffn_block": {"achieved": round(moe_flop / moe_dt / 1e12, 2), "us": round(moe_dt * 1e6, 1),
                                           "frac": round(moe_flop / moe_dt / PEAK[a.precision], 4),
                                           "rows": 2 * B * T, "timed_with": "HIP events on the launch stream"}},
        }
        if dom is not None:
            line["roofline"]["dominant_kernel_alone"] = {
                "name": "fused_mlp_stream_kernel (expert W1-GELU-W2, csrc/mlp_stream.hip)", "achieved": round(dom[1] / dom[0] / 1e12, 2),
                "us": round(dom[0] * 1e6, 1), "frac": round(dom[1] / dom[0] / PEAK[a.precision], 4),
                "flop_per_launch": dom[1], "timed_with": "HIP events on the launch stream around 20 back-to-back launches, kernel alone, "
                                                   "balanced routing, 16-bit output as in the model"}
        if live is not None:
            # the contract's roofline entry: the dominant kernel (26 % of the step, profiles/r01_kernel_stats.txt), live
            pmc_k = None
            if pmc_j is not None:
                per = pmc_j["per_kernel_MB_per_call"]
                per = per.items() if isinstance(per, dict) else per
                pk = next((v for k, v in per if str(k).startswith("fused_mlp_stream_kernel") and ", 7, 4, 512, 0>" in str(k)), None)
                pmc_k = (pk["fetch_x2"] + pk["write"]) * 1e6 if pk else None
            rf = line["roofline"]
            x3 = a.precision == 3  # the fp32-grade mode runs the expert MLP as two grouped bf16x3 GEMMs; the probe brackets the pair
            kname = ("gemm_stream3_kernel<7, 512 | 1024, GELU | none> x 2 (the expert MLP as its two grouped, streamed-weight bf16x3 GEMMs, "
                     "csrc/gemm_stream3.hip)" if x3
                     else "fused_mlp_stream_kernel (expert W1-GELU-W2, csrc/mlp_stream.hip)")
            rf.update({"achieved": round(live[1] / live[0] / 1e12, 2), "frac": round(live[1] / live[0] / PEAK[a.precision], 4),
                       "traffic": None if x3 else pmc_k,
                       "traffic_note": ("per launch of this kernel (mean over its launches in a step); " + traffic_note) if (pmc_k and not x3) else traffic_note,
                       "what": f"dominant kernel {kname}: mean algorithmic FLOP per "
                               "launch (4 * routed rows * D * F) / mean launch duration over the launches of real sampling "
                               "steps; the whole step is under whole_step",
                       "kernel": kname.split(" (")[0], "launch_us_mean": round(live[0] * 1e6, 1),
                       "flop_per_launch_mean": live[1], "launches_timed": len(live[2]),
                       "launches": [{"rows": rw, "us": round(u, 1)} for rw, u in live[2][:8]],
                       "timed_with": "HIP events recorded by the library on the launch stream around every launch of the "
                                     "kernel (mdm_probe_*), eager steps on the live sampler state after the timed region"})
        if modes is not None:
            line["parity_mode"] = modes.pop("parity_mode")
            line["modes"] = modes
            # what `value` is measured in, next to it: only precision 3 claims the reference's 1e-3 bound (parity_mode above)
            mine = next((v for v in modes.values() if v.get("precision") == a.precision), None)
            if mine is not None:
                line["config"]["mode"] = (f"precision {a.precision} ({DTYPE[a.precision]}): median frame error "
                                          f"{mine['median_frame_err_vs_parity_mode']:.1e} vs the parity-grade mode, "
                                          f"{mine['routing_decisions_that_differ']} of {mine['routing_decisions']} routing decisions "
                                          f"differ; the same workload at reference-grade results runs at "
                                          f"{line['parity_mode']['steps_per_s']} steps/s (parity_mode)")
        elif a.precision == 3:
            line["config"]["mode"] = "precision 3 (fp32-grade, the parity mode)"
        if world == 1 and not a.no_other_configs and (a.config, a.sampler) == ("small", "cfg"):
            # the other single-GPU configs of BASELINE.json, measured in this process so that the driver's record carries them
            del r
            torch.cuda.empty_cache()
            line["configs"] = {"configs[2]": other_config_line("big", "cfg", a.schedule, a, dev),
                               "configs[3] (per GPU)": other_config_line("big", "ddim", 100, a, dev)}
            if a.precision in (1, 2):  # configs[4]: B = 64 over 8 GPUs = 8 per GPU, 16 experts, fp8 expert GEMMs; the f16 step beside it
                c4 = other_config_line("big16", "cfg", a.schedule, a, dev, batch=8, precision=5)
                c4["f16_ms_per_step"] = other_config_line("big16", "cfg", a.schedule, a, dev, batch=8, precision=2)["ms_per_step"]
                c4["note"] = ("e4m3 x e4m3 expert GEMMs at this shape: 8.8e-2 median frame error with the oracle's routing imposed, 29 % "
                              "of the routing decisions differ with free routing (fp16 beside it: 2.0e-3, 1.4 %; "
                              "tests/test_fp8_gpu.py); a throughput figure, not a usable-accuracy mode")
                line["configs"]["configs[4] (per GPU)"] = c4
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(host, inputs, a.schedule, a.cfg_scale)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
