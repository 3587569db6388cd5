"""GPU: the path at BASELINE.json's sizes.

* configs[1] (small, 8 experts, B=32, T=196, L=4): the oracle cannot run the whole batch in seconds, so parity is shown
  through properties that do not depend on the size -- samples never interact, so (a) one sample of the
  B=32 HIP forward must equal the oracle run on that sample ALONE, (b) perturbing one sample must leave every other
  sample bit-identical, (c) the cond|uncond rows batched as 2B must equal two separate B-row forwards, (d) repeated
  forwards are bit-identical, (e) the router's counters account for every token.
* configs[0] (small, 4 experts, B=2, T=64, 50-step CFG DDPM, fp32): small enough to run end to end against the oracle.
"""
import os
import sys

import pytest
import torch

from conftest import ROOT, pkg, rel_inf

sys.path.insert(0, os.path.join(ROOT, "oracle"))
import denoiser_ref as R  # noqa: E402
import diffusion_ref as DR  # noqa: E402

pytestmark = pytest.mark.gpu


def _build(E, B, T, precision, N=28, seed=0):
    T_ = pkg("transformer")
    synth = pkg("synth")
    m = T_.MotionTransformer(263, num_frames=196, latent_dim=512, ff_size=1024, num_layers=4, num_heads=4,
                             text_latent_dim=256, moe_num_experts=E, model_size="small", precision=precision)
    sd = synth.synth_state_dict(m._layout, seed)
    m.load_state_dict(sd, strict=True)
    eph = synth.synth_ephemerals(512, 256, 4, 7)
    proj = synth.synth_projections(128, 4, 7)
    m.set_ephemerals(eph), m.set_projections(proj)
    x, _, length, xf_proj, xf_out = synth.synth_inputs(B, T, 263, N, 256, seed, min_len=40)
    host = dict(sd=sd, eph={n: (w, b) for n, w, b in eph}, proj=dict(proj),
                mcfg=dict(latent_dim=512, num_heads=4, num_layers=4, moe_num_experts=E))
    return m.cuda().eval(), host, (x, length, xf_proj, xf_out)


def _oracle(host, x, t, length, xf_proj, xf_out):
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    with torch.no_grad():
        return R.denoiser_forward(host["sd"], host["mcfg"], x, t, length, xf_proj, xf_out, host["eph"], host["proj"])


@pytest.mark.parametrize("precision,tol", [(3, 1e-3)])  # the other modes: error + flip budgets in test_round2_gpu.py
def test_configs1_one_sample_of_the_full_batch_matches_the_oracle(precision, tol):
    B, T = 32, 196
    m, host, (x, length, xf_proj, xf_out) = _build(8, B, T, precision)
    t = torch.full((B,), 977, dtype=torch.int64)
    y = m(x.cuda(), t.cuda(), length.cuda(), xf_proj=xf_proj.cuda(), xf_out=xf_out.cuda()).cpu()
    assert y.shape == (B, T, 263) and torch.isfinite(y).all()
    for b in (0, 17):  # a full-length sample and a ragged one
        ref = _oracle(host, x[b:b + 1], t[b:b + 1], length[b:b + 1], xf_proj[b:b + 1], xf_out[b:b + 1])
        err = rel_inf(y[b:b + 1], ref)
        d = (y[b] - ref[0]).double()
        l2 = float(d.norm() / ref[0].double().norm())
        tok = d.abs().amax(-1) / ref.abs().max()
        print(f"precision {precision} sample {b} (length {int(length[b])}): rel err max {err:.2e}  l2 {l2:.2e}  "
              f"frames with err > 0.05: {int((tok > 0.05).sum())}/{T}  median frame err {float(tok.median()):.2e}")
        if tol is not None:
            assert err < tol, (b, err)


def test_configs1_bf16_mode_error_and_routing_flips():
    """Throughput mode at full depth (8 layers x 2 branches of top-2 routing).  bf16 operand rounding moves gate logits by
    ~1e-3, so some tokens near a routing tie take a different expert than in the fp32 oracle -- an O(1) local change that
    then propagates.  Reported here (bf16 runs report their error and flip count): with the oracle's routing
    injected the error is the arithmetic one (<= 5e-2 of the output range); with free routing the flip fraction is small."""
    T = 196
    m, host, (x, length, xf_proj, xf_out) = _build(8, 32, T, 1)
    b = 17
    sl = slice(b, b + 1)
    t = torch.full((1,), 977, dtype=torch.int64)
    trace = {}
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    with torch.no_grad():
        ref = R.denoiser_forward(host["sd"], host["mcfg"], x[sl], t, length[sl], xf_proj[sl], xf_out[sl], host["eph"],
                                 host["proj"], None, trace)
    names = [f"decoder_blocks_{s}.{i}.module" for s in ("low", "high") for i in range(4)]
    forced = torch.zeros((8, 2 * 2 * T), dtype=torch.int32)
    for li, name in enumerate(names):
        idx = torch.stack([trace[f"{name}.ffn.branches.{br}.top2_idx"] for br in range(2)])  # (2, M, 2)
        forced[li, :idx.numel()] = idx.reshape(-1).to(torch.int32)
    args = (x[sl].cuda(), t.cuda(), length[sl].cuda())
    kw = dict(xf_proj=xf_proj[sl].cuda(), xf_out=xf_out[sl].cuda())
    y_forced = m(*args, forced_routing=forced, **kw).cpu()
    e_forced = rel_inf(y_forced, ref)
    y_free, tr = m(*args, trace=True, **kw)
    e_free = rel_inf(y_free.cpu(), ref)
    d = (y_free.cpu()[0] - ref[0]).abs().amax(-1) / ref.abs().max()
    print(f"bf16 mode, one sample at T={T}: rel err {e_forced:.2e} with the oracle's routing, {e_free:.2e} free "
          f"(frames off by > 5 %: {int((d > 0.05).sum())}/{T}, median frame error {float(d.median()):.2e})")
    assert e_forced < 5e-2
    assert float(d.median()) < 5e-2


def test_configs1_size_independent_properties():
    B, T = 32, 196
    m, host, (x, length, xf_proj, xf_out) = _build(8, B, T, 1)
    dev = "cuda"
    t = torch.full((B,), 500, dtype=torch.int64, device=dev)
    args = dict(xf_proj=xf_proj.to(dev), xf_out=xf_out.to(dev))
    xd, ld = x.to(dev), length.to(dev)
    m.reset_all_moe_counters(m)
    y1 = m(xd, t, ld, **args)
    usage = {k: v.clone() for k, v in m.moe_buffers().items() if k.endswith("expert_usage")}
    y2 = m(xd, t, ld, **args)
    assert torch.equal(y1, y2)                                           # (d) deterministic, run to run
    # (e) every token is counted once per branch (top-1 usage): low-scale layers see B*T/2 tokens, high-scale B*T
    for k, v in usage.items():
        want = B * T // 2 if "decoder_blocks_low" in k else B * T
        assert int(v.sum()) == want, (k, float(v.sum()), want)
    # (b) samples never interact
    xp = xd.clone()
    xp[5] += 0.5
    y3 = m(xp, t, ld, **args)
    others = [b for b in range(B) if b != 5]
    assert torch.equal(y3[others], y1[others]) and not torch.equal(y3[5], y1[5])
    # (c) cond | uncond batched as 2B rows == two B-row forwards (what the captured sampling step relies on)
    xu_p, xu_o = m.uncond_embedding(B, dev) if m._uncond is not None else (args["xf_proj"].flip(0), args["xf_out"].flip(0))
    both = m(torch.cat([xd, xd]), torch.cat([t, t]), torch.cat([ld, ld]), xf_proj=torch.cat([args["xf_proj"], xu_p]),
             xf_out=torch.cat([args["xf_out"], xu_o]))
    sep_u = m(xd, t, ld, xf_proj=xu_p, xf_out=xu_o)
    assert torch.equal(both[:B], y1) and torch.equal(both[B:], sep_u)


def test_configs0_end_to_end_against_the_oracle():
    """BASELINE configs[0]: small / 4 experts / B=2 / T=64 / 50-step DDPM with CFG, fp32-grade mode.  The oracle runs the
    whole 50-step loop (100 forwards of B=2).  EVERY guided step of the HIP sampler is checked from the ORACLE's state
    (teacher forcing) and the routing of that step is dumped (mdm_route_dump) and compared with the oracle's: a step WITHOUT
    a differing decision must match at <= 1e-3; a step with one is a near-tie resolved the other way by two fp32
    implementations (an O(1) local change, not an error) and is accepted only if every differing decision of the FIRST layer
    that has one sits at a token whose oracle margin p2 - p3 is < 1e-5 and there are at most 2 of them (decisions that differ
    in later layers are consequences of that one); at most 3 of the 50 steps may contain such a tie.  The gate therefore does not depend on
    how the compiler associates the gate logits' sums.  The free-running loops are compared for their first two steps."""
    import ctypes as C
    B, T, steps, scale, L = 2, 64, 50, 7.5, 4
    m, host, (x, length, xf_proj, xf_out) = _build(4, B, T, 3, seed=3)
    synth = pkg("synth")
    D = pkg("diffusion")
    lib = pkg("_lib").lib()
    xo_u = synth.uniform_pm1((1, 28, 256), "in.uncond", 3) * (3.0 ** 0.5)
    xp_u = xo_u.mean(1)
    m.set_uncond_embedding(xp_u.cuda(), xo_u.cuda())
    diff = D.GaussianDiffusion(betas=D.get_named_beta_schedule("linear", steps), model_mean_type=D.ModelMeanType.EPSILON,
                               model_var_type=D.ModelVarType.FIXED_SMALL, loss_type=D.LossType.MSE)
    noises = [synth.uniform_pm1((B, T, 263), f"noise.c0.{i}", 3) * (3.0 ** 0.5) for i in range(steps)]
    traj = {}
    kw = {"xf_proj": xf_proj.cuda(), "xf_out": xf_out.cuda(), "length": length.cuda(), "text": ["x"] * B}
    y = diff.p_sample_loop_with_cfg(m, (B, T, 263), noise=x.cuda(), clip_denoised=False, model_kwargs=kw, cfg_scale=scale,
                                    step_noise=noises, callback=lambda i, t, xx: traj.__setitem__(i, xx.clone().cpu()))
    assert torch.isfinite(y).all()
    tb = DR.Tables(DR.linear_betas(steps))
    names = [f"decoder_blocks_{s}.{i}.module" for s in ("low", "high") for i in range(L)]
    torch.set_num_threads(min(16, os.cpu_count() or 1))

    def model(xx, tt, cond, trace):
        xp, xo = (xf_proj, xf_out) if cond else (xp_u.expand(B, -1), xo_u.expand(B, -1, -1))
        with torch.no_grad():
            return R.denoiser_forward(host["sd"], host["mcfg"], xx, tt, length, xp, xo, host["eph"], host["proj"], None, trace)

    xs = x.clone()
    forced, free_errs = [], []
    dump = torch.full((2 * L, 2, 2 * B * T, 2), -1, dtype=torch.int32, device="cuda")
    for i in range(steps):
        t = steps - 1 - i
        tt = torch.full((B,), t, dtype=torch.int64)
        tr_c, tr_u = {}, {}
        nxt, _ = DR.cfg_step(tb, t, xs, model(xs, tt, True, tr_c), model(xs, tt, False, tr_u), noises[i], scale)
        if True:  # EVERY step (no sampling of the gate): one HIP step from the oracle's state, its routing dumped
            lib.mdm_route_dump(C.c_void_p(dump.data_ptr()), C.c_int64(dump.numel()))
            try:
                out = diff.p_sample_with_cfg(m, xs.cuda(), tt.cuda(), clip_denoised=False, model_kwargs=kw, cfg_scale=scale,
                                             noise=noises[i].cuda())
                torch.cuda.synchronize()
            finally:
                lib.mdm_route_dump(C.c_void_p(0), C.c_int64(0))
            # Layers run in this order and both branches of a layer see the same input, so the decisions that differ in the
            # FIRST layer with any are the root cause (they must be near-ties); those in later layers are its consequence (a
            # token routed elsewhere is an O(1) change of that layer's output, hence of every later gate input)
            flips, worst_gap, first = 0, 0.0, None
            for li, name in enumerate(names):
                S = T // 2 if li < L else T
                ours = dump[li].reshape(-1)[:4 * 2 * B * S].reshape(2, 2 * B * S, 2).cpu().long().sort(-1).values
                lf, lg = 0, 0.0
                for br in range(2):
                    want = torch.cat([tr_c[f"{name}.ffn.branches.{br}.top2_idx"], tr_u[f"{name}.ffn.branches.{br}.top2_idx"]])
                    gap = torch.cat([tr_c[f"{name}.ffn.branches.{br}.gap23"], tr_u[f"{name}.ffn.branches.{br}.gap23"]])
                    diffm = (ours[br] != want.long().sort(-1).values).any(-1)
                    lf += int(diffm.sum())
                    if diffm.any():
                        lg = max(lg, float(gap[diffm].max()))
                flips += lf
                if lf and first is None:
                    first = (li, lf)
                    worst_gap = lg
            forced.append((i, rel_inf(out["sample"].cpu(), nxt), flips, worst_gap, first))
        free_errs.append(rel_inf(traj[i], nxt))
        xs = nxt
    assert len(forced) == steps
    clean = [e for _, e, f, _, _ in forced if f == 0]
    print(f"teacher-forced: all {steps} steps checked; {len(clean)} without a differing routing decision (worst rel err "
          f"{max(clean):.1e}); steps with one (step, rel err, decisions that differ, oracle p2-p3 at the first layer's, (first layer, count there)):",
          [(i, f"{e:.1e}", f, f"{gmax:.1e}", fl) for i, e, f, gmax, fl in forced if f])
    print("free-running loop divergence at steps 0, 1, 9, 24, 49:", [f"{free_errs[i]:.1e}" for i in (0, 1, 9, 24, 49)])
    for i, e, f, gmax, fl in forced:
        if f == 0:
            assert e < 1e-3, (i, e)
        else:  # the root-cause decisions: at most 2, each a near-tie of the oracle's own probabilities
            assert fl[1] <= 2 and gmax < 1e-5, (i, e, f, gmax, fl)
    assert len(clean) >= steps - 3
