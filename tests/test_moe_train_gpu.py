"""GPU: training step of the MoE feed-forward block through the C ABI
(mdm_moe_ffn_train_forward / _backward, mdm_sumsq, mdm_adam_step) against torch autograd through the oracle's forward
(oracle/moe_train_ref.py).  Tolerance: gradients within 2e-3 of the largest entry of each tensor (the GEMMs are bf16x3,
the reference arithmetic is fp32); the routing decisions must equal the oracle's."""
import pytest
import torch

from conftest import pkg, rel_inf
from oracle import denoiser_ref as R
from oracle import moe_train_ref as T

pytestmark = pytest.mark.gpu
PREFIX = "blk.ffn"


def _make_sd(D, F, E, Te, seed):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s, scale=1.0: (torch.rand(*s, generator=g) * 2 - 1) * scale
    sd = {}
    for b in range(2):
        p = f"{PREFIX}.branches.{b}"
        sd[p + ".layernorm.weight"], sd[p + ".layernorm.bias"] = 1 + r(D, scale=0.2), r(D, scale=0.1)
        sd[p + ".moe.gate.weight"], sd[p + ".moe.gate.bias"] = r(E, D, scale=0.5), r(E, scale=0.1)   # trained gate: not the zero init
        for e in range(E):
            sd[f"{p}.moe.experts.{e}.0.weight"], sd[f"{p}.moe.experts.{e}.0.bias"] = r(F, D, scale=D ** -0.5), r(F, scale=0.1)
            sd[f"{p}.moe.experts.{e}.2.weight"], sd[f"{p}.moe.experts.{e}.2.bias"] = r(D, F, scale=F ** -0.5), r(D, scale=0.1)
    p = PREFIX + ".proj_out"
    sd[p + ".emb_layers.1.weight"], sd[p + ".emb_layers.1.bias"] = r(2 * D, Te, scale=Te ** -0.5), r(2 * D, scale=0.1)
    sd[p + ".norm.weight"], sd[p + ".norm.bias"] = 1 + r(D, scale=0.2), r(D, scale=0.1)
    sd[p + ".out_layers.2.weight"], sd[p + ".out_layers.2.bias"] = r(D, D, scale=D ** -0.5), r(D, scale=0.1)  # not the zero_module init
    return sd


def _inputs(B, S, D, De, Te, seed):
    g = torch.Generator().manual_seed(seed + 100)
    r = lambda *s, scale=1.0: (torch.rand(*s, generator=g) * 2 - 1) * scale
    eph = None if De == Te else (r(Te, De, scale=De ** -0.5), r(Te, scale=0.1))
    return r(B, S, D, scale=1.5), r(B, De), eph, r(B, S, D)


def _trainer(D, F, E, Te, sd, **kw):
    tr = pkg("moe_train").MoEFFNTrainer(D, F, E, Te, device="cuda", **kw)
    tr.load_state_dict({k: v.cuda() for k, v in sd.items()}, PREFIX)
    return tr


def _hip_forward_backward(tr, x, emb, eph, dout):
    B, S, _ = x.shape
    route = torch.zeros(2, B * S, 2, dtype=torch.int32, device="cuda")
    ephd = None if eph is None else (eph[0].cuda(), eph[1].cuda())
    out = tr.forward(x.cuda(), emb.cuda(), ephd, route_out=route)
    dx, demb = tr.backward(dout.cuda())
    return out.cpu(), dx.cpu(), demb.cpu(), route.cpu().long(), tr.lb_loss.cpu().clone()


@pytest.mark.parametrize("D,F,E,Te,De,B,S", [(128, 256, 4, 96, 64, 3, 10), (512, 1024, 8, 2048, 512, 4, 24), (64, 128, 3, 64, 64, 2, 7),
                                            (256, 320, 16, 128, 256, 2, 33)])
def test_block_gradients_match_autograd(D, F, E, Te, De, B, S):
    sd = _make_sd(D, F, E, Te, seed=D + E)
    x, emb, eph, dout = _inputs(B, S, D, De, Te, seed=S)
    tr = _trainer(D, F, E, Te, sd)
    out, dx, demb, route, lb = _hip_forward_backward(tr, x, emb, eph, dout)
    # free routing of the oracle equals the HIP router's decisions; gradients are then compared on that routing
    o_out, o_dx, o_demb, o_g, o_lb, trace = T.moe_ffn_grads(sd, PREFIX, E, x, emb, eph, dout)
    for b in range(2):
        assert torch.equal(trace[f"{PREFIX}.branches.{b}.top2_idx"], route[b]), f"branch {b}: routing differs from the oracle"
    errs = {"out": rel_inf(out, o_out), "dx": rel_inf(dx, o_dx), "demb": rel_inf(demb, o_demb), "lb_loss": rel_inf(lb, o_lb)}
    keys = pkg("moe_train").reference_keys(PREFIX, E)
    untouched = 0
    for name, ks in keys.items():
        g = tr.grads.views[name].cpu()
        ref = torch.stack([o_g.get(k, torch.zeros_like(sd[k])) for k in ks]).reshape(g.shape)
        untouched += sum(k not in o_g for k in ks)
        errs["d" + name] = rel_inf(g, ref)
    print(f"MoE block training D={D} F={F} E={E} B*S={B * S}: " + ", ".join(f"{k} {v:.1e}" for k, v in errs.items()),
          f"({untouched} parameter tensors of unused experts: zero gradient)")
    assert max(errs.values()) < 2e-3, errs


@pytest.mark.parametrize("D,F,E,Te,De,B,S,p", [(256, 320, 4, 128, 256, 2, 19, 0.1), (512, 1024, 8, 2048, 512, 2, 24, 0.1), (256, 128, 3, 96, 64, 1, 9, 0.5)])
def test_training_mode_dropout_matches_the_oracle_with_the_same_masks(D, F, E, Te, De, B, S, p):
    """Dropout on each branch's output (multi_branch.py:57) and after the SiLU of the stylization block (stylization.py:16):
    the counter-based masks are restated in numpy (oracle/moe_train_ref.dropout_masks) and applied to the oracle's forward;
    outputs and every gradient must then agree as in eval mode, and the fraction of dropped elements must be p."""
    sd = _make_sd(D, F, E, Te, seed=D + E + 1)
    x, emb, eph, dout = _inputs(B, S, D, De, Te, seed=S + 1)
    seed = 0x1234567800000007
    tr = _trainer(D, F, E, Te, sd, dropout=p, seed=seed)
    out, dx, demb, route, lb = _hip_forward_backward(tr, x, emb, eph, dout)
    masks = T.dropout_masks(seed, B * S, D, p)
    frac = float(sum((m == 0).float().mean() for m in masks) / 3)
    assert abs(frac - p) < 0.03, frac
    o_out, o_dx, o_demb, o_g, o_lb, trace = T.moe_ffn_grads(sd, PREFIX, E, x, emb, eph, dout, masks=masks)
    for b in range(2):
        assert torch.equal(trace[f"{PREFIX}.branches.{b}.top2_idx"], route[b])
    errs = {"out": rel_inf(out, o_out), "dx": rel_inf(dx, o_dx), "demb": rel_inf(demb, o_demb)}
    for name, ks in pkg("moe_train").reference_keys(PREFIX, E).items():
        g = tr.grads.views[name].cpu()
        errs["d" + name] = rel_inf(g, torch.stack([o_g.get(k, torch.zeros_like(sd[k])) for k in ks]).reshape(g.shape))
    print(f"training-mode dropout p={p} D={D}: dropped {frac:.3f}; " + ", ".join(f"{k} {v:.1e}" for k, v in errs.items()))
    assert max(errs.values()) < 2e-3, errs
    # eval behaviour is a different forward, and a different seed is a different mask
    tr0 = _trainer(D, F, E, Te, sd)
    assert rel_inf(tr0.forward(x.cuda(), emb.cuda(), None if eph is None else (eph[0].cuda(), eph[1].cuda())).cpu(), out) > 1e-2
    with pytest.raises(Exception):
        pkg("moe_train").MoEFFNTrainer(64, 64, 4, 64, dropout=0.1).forward(torch.zeros(1, 4, 64, device="cuda"), torch.zeros(1, 64, device="cuda"))


def test_unused_experts_get_zero_gradients_and_empty_groups_are_safe():
    """E = 16 with 5 tokens: most expert groups are empty (K range of length 0 in the weight-gradient GEMM)."""
    D, F, E, Te = 64, 64, 16, 64
    sd = _make_sd(D, F, E, Te, seed=3)
    x, emb, eph, dout = _inputs(1, 5, D, Te, Te, seed=9)
    tr = _trainer(D, F, E, Te, sd)
    for v in tr.grads.views.values():
        v.fill_(float("nan"))  # the backward overwrites every gradient
    out, dx, demb, route, lb = _hip_forward_backward(tr, x, emb, eph, dout)
    assert all(bool(torch.isfinite(v).all()) for v in tr.grads.views.values())
    used = [set(route[b].flatten().tolist()) for b in range(2)]
    w1 = tr.grads.views["w1"].cpu()
    for b in range(2):
        for e in range(E):
            assert (w1[b, e].abs().max() > 0) == (e in used[b])


def test_clip_and_adam_match_torch():
    D, F, E, Te = 64, 128, 4, 64
    tr = _trainer(D, F, E, Te, _make_sd(D, F, E, Te, seed=1), lr=1e-3, max_norm=1.0)
    g = torch.Generator().manual_seed(5)
    p0 = tr.params.flat.cpu().clone()
    m = torch.zeros_like(p0)
    v = torch.zeros_like(p0)
    p = p0.clone()
    for step in range(1, 4):
        grads = (torch.rand(p0.numel(), generator=g) * 2 - 1) * (0.5 if step == 2 else 1e-4)  # clipped and unclipped steps
        tr.grads.flat.copy_(grads)
        tr.optimizer_step()
        p, m, v, norm = T.adam_clip_step(p, grads, m, v, step, lr=1e-3, max_norm=1.0)
        assert abs(tr.grad_norm() - float(norm)) <= 1e-5 * float(norm)
        assert rel_inf(tr.params.flat.cpu() - p0, p - p0) < 2e-4  # the updates are ~1e-3 of fp32 parameters of order 1
        assert rel_inf(tr.adam_v.cpu(), v) < 1e-4  # v ~ (clip * g)^2: twice the fp32 rounding of the norm


def test_three_training_steps_follow_a_torch_training_loop():
    """forward -> masked MSE -> backward -> clip -> Adam, three iterations, against the same loop written with autograd and
    torch.optim.Adam over the oracle's forward (ddpm_trainer.py:201-244)."""
    D, F, E, Te, De, B, S = 128, 256, 4, 96, 64, 2, 12
    sd = _make_sd(D, F, E, Te, seed=11)
    x, emb, eph, target = _inputs(B, S, D, De, Te, seed=4)
    mask = torch.ones(B, S)
    mask[1, 8:] = 0
    tr = _trainer(D, F, E, Te, sd, lr=1e-3)
    ref = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    opt = torch.optim.Adam(list(ref.values()), lr=1e-3)
    for it in range(3):
        logs = tr.train_step(x.cuda(), emb.cuda(), target.cuda(), (eph[0].cuda(), eph[1].cuda()), mask.cuda())
        opt.zero_grad()
        trace = {}
        out = R.moe_ffn(x, emb, ref, PREFIX, E, eph, trace=trace)
        loss = (((out - target) ** 2).mean(-1) * mask).sum() / mask.sum()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(list(ref.values()), 1.0)
        opt.step()
        assert abs(logs["loss_mot_rec"] - float(loss)) < 2e-4 * max(1.0, abs(float(loss))), (it, logs, float(loss))
        # the reference logs the unscaled load-balancing sum and adds it unscaled (ddpm_trainer.py:217-222)
        assert logs["loss_moe"] > 0 and abs(logs["loss_total"] - (logs["loss_mot_rec"] + logs["loss_moe"])) < 1e-6
        assert abs(logs["loss_moe_scaled"] - 0.01 * logs["loss_moe"]) < 1e-9
    new = tr.state_dict(PREFIX)
    # Adam divides by sqrt(v): an entry whose gradient is at the rounding level moves by +-lr whatever its value, so single
    # entries are ill-conditioned; the update VECTOR is compared in the L2 sense, tensor by tensor
    worst = 0.0
    for k in new:
        du, dr = new[k].cpu() - sd[k], ref[k].detach() - sd[k]
        if float(dr.norm()) > 0:
            worst = max(worst, float((du - dr).norm() / dr.norm()))
        else:
            assert float(du.abs().max()) == 0.0, k  # experts nobody routed to: untouched on both sides
    print(f"parameter updates after 3 steps vs torch loop: worst relative L2 error {worst:.2e}")
    assert worst < 5e-3
