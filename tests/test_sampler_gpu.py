"""GPU parity of the sampling loops (HIP step kernels + graph-captured forward) against the reference's own loops."""
import types

import pytest
import torch

from conftest import build_module, load_golden, rel_inf, pkg

pytestmark = pytest.mark.gpu


def _setup():
    g, meta = load_golden("loops_tiny")
    m, _ = build_module(meta, precision=3)
    D = pkg("diffusion")
    synth = pkg("synth")
    B, T, F_ = g["x_T"].shape

    def noises(tag, n):
        return [synth.uniform_pm1((B, T, F_), f"noise.{tag}.{i}", meta["iseed"]) * (3.0 ** 0.5) for i in range(n)]

    def diff(steps):
        return D.GaussianDiffusion(betas=D.get_named_beta_schedule("linear", steps), model_mean_type=D.ModelMeanType.EPSILON,
                                   model_var_type=D.ModelVarType.FIXED_SMALL, loss_type=D.LossType.MSE)

    kw = {"xf_proj": g["xf_proj"].cuda(), "xf_out": g["xf_out"].cuda(), "length": g["length"].cuda(),
          "text": ["a person walks"] * B}
    m.set_uncond_embedding(g["xf_proj_uncond"][:1].cuda(), g["xf_out_uncond"][:1].cuda())
    return g, meta, m, diff, noises, kw


@pytest.mark.parametrize("use_graph", [True, False])
def test_cfg_loop_matches_reference(use_graph):
    g, meta, m, diff, noises, kw = _setup()
    d = diff(meta["steps_cfg"])
    traj = {}
    want = g["cfg/traj_idx"].tolist()
    y = d.p_sample_loop_with_cfg(m, tuple(g["x_T"].shape), noise=g["x_T"].cuda(), clip_denoised=False, model_kwargs=kw,
                                 cfg_scale=meta["cfg_scale"], step_noise=noises("cfg", meta["steps_cfg"]),
                                 use_graph=use_graph,
                                 callback=lambda i, t, x: traj.__setitem__(i, x.clone().cpu()) if i in want else None)
    for j, i in enumerate(want):
        assert rel_inf(traj[i], g["cfg/traj"][j]) < 1e-3, i
    assert rel_inf(y.cpu(), g["cfg/final"]) < 1e-3


@pytest.mark.parametrize("eta", [0.0, 0.5])
def test_ddim_loop_matches_reference(eta):
    g, meta, m, diff, noises, kw = _setup()
    d = diff(meta["steps_ddim"])
    kw2 = {k: kw[k] for k in ("xf_proj", "xf_out", "length")}
    y = d.ddim_sample_loop(m, tuple(g["x_T"].shape), noise=g["x_T"].cuda(), model_kwargs=kw2, eta=eta,
                           step_noise=noises(f"ddim.{eta}", meta["steps_ddim"]))
    assert rel_inf(y.cpu(), g[f"ddim{eta}/final"]) < 1e-3


def test_trainer_generate_api():
    """DDPMTrainer.generate (ddpm_trainer.py:176-199): list of (T, dim_pose) tensors, batches of batch_size."""
    g, meta, m, diff, noises, kw = _setup()
    Tr = pkg("trainer")
    m.text_encoder_fn = lambda text, device: (g["xf_proj"][:1].expand(len(text), -1).to(device),
                                              g["xf_out"][:1].expand(len(text), -1, -1).to(device))
    args = types.SimpleNamespace(device=torch.device("cuda"), diffusion_steps=25, is_train=False, cfg_scale=2.5)
    tr = Tr.DDPMTrainer(args, m)
    outs = tr.generate(["a", "b", "c"], torch.tensor([16, 12, 16]), 263, batch_size=2)
    assert len(outs) == 3 and all(o.shape == (16, 263) for o in outs)
    assert all(torch.isfinite(o).all() for o in outs)
