"""CPU: host logic of the MoE-block training step -- the oracle's gradients against fp64 finite differences, the
state_dict key map against the reference's layout dump, and the gradient all-reduce on gloo (world size 2)."""
import json
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import pkg
from oracle import moe_train_ref as T

HERE = os.path.dirname(os.path.abspath(__file__))


def test_reference_keys_cover_the_block_of_the_reference_layout():
    layout = dict(json.load(open(os.path.join(HERE, "golden", "state_dict_layout.json")))["small_E8_L4"]["keys"])
    prefix = "decoder_blocks_low.0.module.ffn"
    want = {k for k in layout if k.startswith(prefix + ".") and "expert_usage" not in k and "expert_importance" not in k}
    keys = pkg("moe_train").reference_keys(prefix, 8)
    got = {k for ks in keys.values() for k in ks}
    assert got == want
    shapes = dict(pkg("moe_train")._shapes(512, 1024, 8, 2048))
    for name, ks in keys.items():
        per = list(shapes[name][(0 if len(ks) == 1 else 1 if len(ks) == 2 else 2):])
        assert all(layout[k] == per for k in ks), name


def test_oracle_gradients_match_finite_differences():
    from test_moe_train_gpu import _make_sd, _inputs, PREFIX
    D, F, E, Te = 16, 24, 3, 16
    sd = {k: v.double() for k, v in _make_sd(D, F, E, Te, seed=2).items()}
    x, emb, eph, dout = [None if t is None else (t.double() if torch.is_tensor(t) else t) for t in _inputs(2, 5, D, Te, Te, seed=1)]
    out, dx, demb, grads, lb, trace = T.moe_ffn_grads(sd, PREFIX, E, x, emb, None, dout, dtype=torch.float64)
    forced = [trace[f"{PREFIX}.branches.{b}.top2_idx"] for b in range(2)]
    from oracle import denoiser_ref as R
    f = lambda s, xx: float((R.moe_ffn(xx, emb, s, PREFIX, E, None, forced=forced) * dout).sum())
    h = 1e-6
    for key in (f"{PREFIX}.branches.1.moe.gate.weight", f"{PREFIX}.branches.0.moe.experts.{int(forced[0][0, 0])}.0.weight",
                f"{PREFIX}.proj_out.norm.weight", f"{PREFIX}.branches.0.layernorm.bias"):
        idx = (0,) * sd[key].dim()
        sp, sm = dict(sd), dict(sd)
        sp[key], sm[key] = sd[key].clone(), sd[key].clone()
        sp[key][idx] += h
        sm[key][idx] -= h
        fd = (f(sp, x) - f(sm, x)) / (2 * h)
        assert abs(fd - float(grads[key][idx])) < 1e-5 * max(1.0, abs(fd)), key
    xp, xm = x.clone(), x.clone()
    xp[0, 0, 0] += h
    xm[0, 0, 0] -= h
    assert abs((f(sd, xp) - f(sd, xm)) / (2 * h) - float(dx[0, 0, 0])) < 1e-5


def _ar_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    flat = torch.arange(10, dtype=torch.float32) * (rank + 1)
    pkg("moe_train").all_reduce_mean_(flat)
    q.put((rank, flat))
    dist.destroy_process_group()


def test_gradient_all_reduce_is_the_mean_over_ranks_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 500
    ps = [ctx.Process(target=_ar_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    res = dict(q.get(timeout=120) for _ in ps)
    [p.join(60) for p in ps]
    want = torch.arange(10, dtype=torch.float32) * 1.5
    assert torch.equal(res[0], want) and torch.equal(res[1], want)


def test_training_step_refuses_cpu():
    with pytest.raises(Exception):
        pkg("moe_train").MoEFFNTrainer(64, 64, 4, 64, device="cpu")


def test_logged_losses_follow_the_reference_trainer():
    """ddpm_trainer.py:217-222 logs loss_moe = the unscaled get_moe_loss (gaussian_diffusion.py:985) and
    loss_total = loss_mot_rec + loss_moe; the 0.01 of get_total_moe_loss (transformer.py:265-270) is not on that path."""
    logs = pkg("moe_train").loss_logs(torch.tensor(0.25), torch.tensor(3.0), moe_coef=0.01)
    assert list(logs)[:3] == ["loss_mot_rec", "loss_moe", "loss_total"]  # backward_G's OrderedDict
    assert logs["loss_mot_rec"] == 0.25 and logs["loss_moe"] == 3.0 and logs["loss_total"] == 3.25
    assert abs(logs["loss_moe_scaled"] - 0.03) < 1e-12
    assert all(isinstance(v, float) for v in logs.values())
