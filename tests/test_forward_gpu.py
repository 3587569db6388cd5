"""GPU parity: the HIP denoiser (through libmdm_hip.so) against golden vectors produced by the reference."""
import pytest
import torch

from conftest import build_module, load_golden, rel_inf, pkg

pytestmark = pytest.mark.gpu

CASES = ["fwd_tiny", "fwd_tiny_l2", "fwd_tiny_eqdim", "fwd_small_dims", "fwd_big_dims", "fwd_tools_shape"]
TOL_FP32 = 1e-3  # north_star: 1e-3 relative vs the reference CPU denoiser


def _layer_names(L):
    return [f"decoder_blocks_{s}.{i}.module" for s in ("low", "high") for i in range(L)]


@pytest.mark.parametrize("case", CASES)
def test_forward_matches_reference_fp32_grade(case):
    g, meta = load_golden(case)
    m, _ = build_module(meta, precision=3)
    dev = "cuda"
    y, tr = m(g["x"].to(dev), g["timesteps"].to(dev), g["length"].to(dev), xf_proj=g["xf_proj"].to(dev),
              xf_out=g["xf_out"].to(dev), trace=True)
    B, T, _ = g["x"].shape
    D, L = meta["latent_dim"], meta["cfg"]["num_layers"]
    errs = {}
    for li, name in enumerate(_layer_names(L)):
        M = B * T // 2 if li < L else B * T
        flat = tr[li].reshape(-1)
        for slot, sub in enumerate(("dual_self_attn", "cross_attn", "ffn", "sd_cross_attn")):
            key = f"trace/{name}.{sub}"
            if key in g:
                ours = flat[slot * M * D:(slot + 1) * M * D].reshape(g[key].shape).cpu()
                errs[f"{name}.{sub}"] = rel_inf(ours, g[key])
    errs["output"] = rel_inf(y.cpu(), g["output"])
    bad = {k: v for k, v in errs.items() if not v < TOL_FP32}
    assert not bad, bad
    print(case, "max rel err", max(errs.values()))
    # MoE counters (switch_moe.py:71-92) are maintained device side
    sd_now = m.state_dict()
    for k, v in g.items():
        if k.startswith("buf/") and k.endswith("expert_usage"):
            assert torch.equal(sd_now[k[4:]].cpu(), v), k
        if k.startswith("buf/") and k.endswith("expert_importance"):
            assert torch.allclose(sd_now[k[4:]].cpu(), v, rtol=1e-4, atol=1e-4), k


@pytest.mark.parametrize("case", ["fwd_tiny", "fwd_small_dims", "fwd_big_dims"])
def test_forward_bf16_reports_error(case):
    """Single-pass bf16 MFMA mode (the throughput mode): bounded error; routing forced to the reference's
    choice so that an expert flip (O(1) local change) does not masquerade as arithmetic error."""
    g, meta = load_golden(case)
    m, _ = build_module(meta, precision=1)
    dev = "cuda"
    B, T, _ = g["x"].shape
    L = meta["cfg"]["num_layers"]
    forced = torch.zeros((2 * L, 2 * 2 * B * T), dtype=torch.int32)
    for li, name in enumerate(_layer_names(L)):
        M = B * T // 2 if li < L else B * T
        idx = torch.stack([g[f"trace/{name}.ffn.branches.{b}.moe.top2_idx"] for b in range(2)])  # (2, M, 2)
        forced[li, :idx.numel()] = idx.reshape(-1).to(torch.int32)
    y = m(g["x"].to(dev), g["timesteps"].to(dev), g["length"].to(dev), xf_proj=g["xf_proj"].to(dev),
          xf_out=g["xf_out"].to(dev), forced_routing=forced)
    err = rel_inf(y.cpu(), g["output"])
    print(case, "bf16 rel err", err)
    assert err < 5e-2


def test_rejects_cpu_and_odd_T():
    g, meta = load_golden("fwd_tiny")
    m, _ = build_module(meta, device="cuda")
    L = pkg("_lib")
    with pytest.raises(L.MdmError):
        m(g["x"], g["timesteps"], g["length"], xf_proj=g["xf_proj"], xf_out=g["xf_out"])
    with pytest.raises(ValueError):
        m(g["x"][:, :15].cuda(), g["timesteps"].cuda(), g["length"].cuda(), xf_proj=g["xf_proj"].cuda(),
          xf_out=g["xf_out"].cuda())


SHAPES = [(1, 2, 1), (1, 6, 5), (3, 30, 32), (2, 62, 33), (5, 14, 28), (2, 196, 85), (33, 4, 8), (4, 100, 31), (1, 196, 28)]


@pytest.mark.parametrize("B,T,N", SHAPES)
def test_forward_shape_sweep_both_modes(B, T, N):
    """Odd shapes through every fused kernel and its fallback (T/2 odd, one frame pair, N at and past the folded
    cross-attention's limit of 32 tokens, more samples than one wave of workgroups): both precision modes against the
    oracle on the real small widths (D=512, 8 experts, one layer per scale), routing injected from the oracle."""
    import os, sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import denoiser_ref as R
    synth = pkg("synth")
    g, meta = load_golden("fwd_small_dims")
    seed = 100 + B + T + N
    x = synth.uniform_pm1((B, T, 263), "sweep.x", seed) * (3.0 ** 0.5)
    xf_out = synth.uniform_pm1((B, N, 256), "sweep.xf", seed) * (3.0 ** 0.5)
    xf_proj = xf_out.mean(1)
    gen = torch.Generator().manual_seed(seed)
    length = torch.randint(1, T + 1, (B,), generator=gen)  # any length, not only multiples of 4
    length[0] = T
    ts = torch.randint(0, 1000, (B,), generator=gen)
    trace = {}
    m3, (sd, eph, proj, mcfg) = build_module(meta, precision=3)
    with torch.no_grad():
        ref = R.denoiser_forward(sd, mcfg, x, ts, length, xf_proj, xf_out, eph, proj, None, trace)
    forced = torch.zeros((2, 2 * 2 * B * T), dtype=torch.int32)
    for li, name in enumerate(_layer_names(1)):
        idx = torch.stack([trace[f"{name}.ffn.branches.{b}.top2_idx"] for b in range(2)])
        forced[li, :idx.numel()] = idx.reshape(-1).to(torch.int32)
    dev = "cuda"
    args = (x.to(dev), ts.to(dev), length.to(dev))
    kw = dict(xf_proj=xf_proj.to(dev), xf_out=xf_out.to(dev))
    e3 = rel_inf(m3(*args, **kw).cpu(), ref)
    m1, _ = build_module(meta, precision=1)
    e1 = rel_inf(m1(*args, forced_routing=forced, **kw).cpu(), ref)
    print(f"B={B} T={T} N={N}: fp32-grade {e3:.2e}  bf16 {e1:.2e}")
    assert e3 < TOL_FP32 and e1 < 5e-2


def test_zero_and_unit_length_samples_match_the_oracle():
    """Degenerate lengths: a sample with NO valid frame (every Performer key masked: the denominator clamp of
    fast_attention.py:82 decides the output), one with a single frame and one odd length whose half-scale length rounds down
    (transformer.py:341-342), next to a full-length sample -- fp32-grade and f16 (routing injected) against the oracle."""
    import os, sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import denoiser_ref as R
    synth = pkg("synth")
    g, meta = load_golden("fwd_small_dims")
    B, T, N = 4, 24, 7
    x = synth.uniform_pm1((B, T, 263), "deg.x", 5) * (3.0 ** 0.5)
    xf_out = synth.uniform_pm1((B, N, 256), "deg.xf", 5) * (3.0 ** 0.5)
    xf_proj = xf_out.mean(1)
    length = torch.tensor([T, 0, 1, 13])
    ts = torch.tensor([999, 0, 500, 17])
    trace = {}
    m3, (sd, eph, proj, mcfg) = build_module(meta, precision=3)
    with torch.no_grad():
        ref = R.denoiser_forward(sd, mcfg, x, ts, length, xf_proj, xf_out, eph, proj, None, trace)
    assert torch.isfinite(ref).all()
    forced = torch.zeros((2, 2 * 2 * B * T), dtype=torch.int32)
    for li, name in enumerate(_layer_names(1)):
        idx = torch.stack([trace[f"{name}.ffn.branches.{b}.top2_idx"] for b in range(2)])
        forced[li, :idx.numel()] = idx.reshape(-1).to(torch.int32)
    args = (x.cuda(), ts.cuda(), length.cuda())
    kw = dict(xf_proj=xf_proj.cuda(), xf_out=xf_out.cuda())
    y3 = m3(*args, **kw).cpu()
    m2, _ = build_module(meta, precision=2)
    y2 = m2(*args, forced_routing=forced, **kw).cpu()
    e3, e2 = rel_inf(y3, ref), rel_inf(y2, ref)
    print(f"lengths {length.tolist()}: fp32-grade {e3:.2e}, f16 (routing injected) {e2:.2e}")
    assert torch.isfinite(y3).all() and torch.isfinite(y2).all()
    assert e3 < TOL_FP32 and e2 < 1e-2


def test_probe_brackets_every_expert_mlp_launch():
    """mdm_probe_enable / mdm_probe_read (bench.py's live timing of the dominant kernel): one event pair per fused expert
    MLP launch of a throughput-mode forward, with that launch's routed-row count."""
    import ctypes as C
    L = pkg("_lib")
    g, meta = load_golden("fwd_small_dims")
    m, _ = build_module(meta, precision=1)
    dev = "cuda"
    args = (g["x"].to(dev), g["timesteps"].to(dev), g["length"].to(dev))
    kw = dict(xf_proj=g["xf_proj"].to(dev), xf_out=g["xf_out"].to(dev))
    m(*args, **kw)
    lib = L.lib()
    L.check(lib.mdm_probe_enable(1))
    m(*args, **kw)
    us, rows = (C.c_float * 8)(), (C.c_int32 * 8)()
    n = lib.mdm_probe_read(us, rows, 8)
    L.check(lib.mdm_probe_enable(0))
    B, T, _ = g["x"].shape
    assert n == 2  # one layer per scale in this golden config
    assert [rows[i] for i in range(n)] == [4 * B * T // 2, 4 * B * T] and all(us[i] > 0 for i in range(n))
    m(*args, **kw)
    assert lib.mdm_probe_read(us, rows, 8) == 2  # disabled: nothing new is recorded
