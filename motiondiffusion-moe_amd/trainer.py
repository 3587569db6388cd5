"""Sampling half of the reference's ``DDPMTrainer`` (text2motion/trainers/ddpm_trainer.py:28-71,145-199,246-289).

Same constructor arguments (``args.device``, ``args.diffusion_steps``, ``args.is_train``, optional ``cfg_scale``),
same ``generate(caption, m_lens, dim_pose, batch_size)`` -> list of (T, dim_pose) tensors, same checkpoint dict keys
(``encoder``, ``ep``, ``total_it``, ``opt_encoder``).  The training loop (forward/backward/update/train) is out of
scope for this build and raises.
"""
from __future__ import annotations

import torch

from .diffusion import GaussianDiffusion, LossType, ModelMeanType, ModelVarType, get_named_beta_schedule


class DDPMTrainer(object):
    def __init__(self, args, encoder):
        self.opt = args
        self.device = args.device
        self.encoder = encoder
        self.diffusion_steps = args.diffusion_steps
        betas = get_named_beta_schedule("linear", self.diffusion_steps)
        self.diffusion = GaussianDiffusion(betas=betas, model_mean_type=ModelMeanType.EPSILON,
                                           model_var_type=ModelVarType.FIXED_SMALL, loss_type=LossType.MSE)
        self.sampler_name = "uniform"
        self.to(self.device)
        self.cfg_scale = getattr(args, "cfg_scale", 7.5)

    def _model(self):
        return self.encoder.module if hasattr(self.encoder, "module") else self.encoder

    def to(self, device):
        self._model().to(device)

    def train_mode(self):
        self._model().train()

    def eval_mode(self):
        self._model().eval()

    @torch.no_grad()
    def generate_batch(self, caption, m_lens, dim_pose, *, noise=None, step_noise=None, progress=True, seed=None,
                       sample_offset=0):
        m = self._model()
        xf_proj, xf_out = m.encode_text(caption, self.device)
        m_lens = torch.as_tensor(m_lens)
        T = min(int(m_lens.max()), m.num_frames)
        B = len(caption)
        return self.diffusion.p_sample_loop_with_cfg(
            m, (B, T, dim_pose), clip_denoised=False, progress=progress, noise=noise, step_noise=step_noise,
            model_kwargs={"xf_proj": xf_proj, "xf_out": xf_out, "length": m_lens, "text": caption},
            cfg_scale=self.cfg_scale, seed=seed, sample_offset=sample_offset)

    @torch.no_grad()
    def generate(self, caption, m_lens, dim_pose, batch_size=8, *, progress=False, seed=None, noises=None):
        """``seed``: sample i's noise is then a function of (seed, i) only (counter-based device generator), so the result
        does not depend on ``batch_size``; without it the torch generator is used, as in the reference.
        ``noises``: optional list with one ``(x_T, [step noise, ...])`` pair per batch, replacing the draws (parity tests)."""
        N = len(caption)
        self.eval_mode()
        all_output = []
        cur = 0
        while cur < N:
            end = min(cur + batch_size, N)
            x_T, step_noise = noises[cur // batch_size] if noises is not None else (None, None)
            out = self.generate_batch(caption[cur:end], m_lens[cur:end], dim_pose, progress=progress, seed=seed,
                                      sample_offset=cur, noise=x_T, step_noise=step_noise)
            all_output.extend(out[i] for i in range(out.shape[0]))
            cur += batch_size
        return all_output

    @torch.no_grad()
    def generate_bucketed(self, caption, m_lens, dim_pose, batch_size=32, *, unit_length=4, seed=None, group=None,
                          progress=False):
        """Evaluation-scale variant of ``generate``: same inputs and the same kind of result (a
        list of per-sample ``(T_batch, dim_pose)`` tensors in the caller's order, valid up to each sample's length),
        but batches hold samples of similar length (less padded work) and, under ``torch.distributed``, are dealt over
        the ranks with one all_gather at the end.  With ``seed`` every sample's noise is a function of (seed, its index in
        ``caption``) only -- the same as ``generate(..., seed=)`` -- so on each sample's valid frames the two give identical
        results whatever the bucketing (tests/test_sampler_gpu.py)."""
        from . import dist as D
        m = self._model()
        self.eval_mode()
        lens = torch.as_tensor(m_lens).flatten().long().cpu()
        plan = D.plan_buckets(lens, batch_size, m.num_frames, unit_length)

        def run_bucket(k, idx, T):
            cap = [caption[i] for i in idx.tolist()]
            ln = lens[idx].clamp(max=T).to(self.device)
            xf_proj, xf_out = m.encode_text(cap, self.device)
            return self.diffusion.p_sample_loop_with_cfg(
                m, (len(cap), T, dim_pose), clip_denoised=False, progress=progress,
                model_kwargs={"xf_proj": xf_proj, "xf_out": xf_out, "length": ln, "text": cap}, cfg_scale=self.cfg_scale,
                seed=seed, sample_offset=idx)  # noise keyed on each row's index in the CALLER's list: == generate(seed=)

        return D.run_plan(plan, run_bucket, len(caption), m.num_frames, dim_pose, self.device, group)

    @torch.no_grad()
    def generate_joints(self, caption, m_lens, dim_pose, mean, std, batch_size=8, *, joints_num=22, sigma=1.0,
                        bucketed=False, **kw):
        """``generate`` followed by the reference's post-processing (tools/visualization.py:21-27,89) on the device:
        list of ``(m_len, joints_num, 3)`` joint positions, temporally smoothed with a gaussian of width ``sigma``."""
        from .postprocess import motion_to_joints
        gen = self.generate_bucketed if bucketed else self.generate
        motions = gen(caption, m_lens, dim_pose, batch_size, **kw)
        lens = [min(int(n), mo.shape[0]) for n, mo in zip(torch.as_tensor(m_lens).flatten().tolist(), motions)]
        x = torch.zeros((len(motions), max(mo.shape[0] for mo in motions), dim_pose), device=motions[0].device)
        for i, mo in enumerate(motions):
            x[i, :mo.shape[0]] = mo
        j = motion_to_joints(x, mean, std, torch.tensor(lens), joints_num, sigma)  # one launch for all samples
        return [j[i, :n] for i, n in enumerate(lens)]

    def save(self, file_name, ep, total_it):
        state = {"opt_encoder": getattr(self, "opt_encoder_state", {}), "ep": ep, "total_it": total_it,
                 "encoder": self._model().state_dict()}
        torch.save(state, file_name)

    def load(self, model_dir):
        ckpt = torch.load(model_dir, map_location=self.device)
        self._model().load_state_dict(ckpt["encoder"], strict=False)
        return ckpt["ep"], ckpt.get("total_it", 0)

    def train(self, *a, **k):
        raise NotImplementedError("whole-model training is outside this build's scope (SURVEY.md section 8(f)): DDPMTrainer "
                                  "provides the sampling API; the training step of the MoE feed-forward block is "
                                  "moe_train.MoEFFNTrainer")

    forward = backward_G = update = train
