"""Training step of one MoE feed-forward block on the HIP path.

Mirrors what the reference does for this block inside a training iteration:
  * forward of ``MoEMultiBranchFFN`` in training mode      text2motion/models/multi_branch.py:52-61,
    ``SwitchMoELayer.forward``                               text2motion/models/switch_moe.py:44-111,
    ``StylizationBlock.forward``                             text2motion/models/stylization.py:20-31
  * ``get_load_balancing_loss``                              text2motion/models/switch_moe.py:113-145
  * ``loss.backward()`` / ``clip_grad_norm_(.., 1.0)`` / ``Adam.step`` of ``DDPMTrainer.update``
                                                             text2motion/trainers/ddpm_trainer.py:228-244
  * the gradient all-reduce that ``DistributedDataParallel`` performs (tools/train.py:140-145): here ONE collective over the
    block's flat gradient buffer (RCCL over xGMI; `gloo` in the CPU tests of the host logic).

Parameters, gradients and the Adam moments live in three flat fp32 device buffers; the per-tensor views follow the
reference's state_dict layouts with the two branches (and the experts) stacked on leading dimensions.  Dropout
(multi_branch.py:57, stylization.py:16) uses counter-based masks keyed on (seed, site, row, element): ``dropout=p`` with a
fresh ``seed`` per iteration (the trainer advances it itself) reproduces a training-mode forward; ``dropout=0`` is eval
behaviour.  There is no CPU fallback: without the HIP library every call raises.
"""
import ctypes as C
import math
from typing import Dict, Optional, Tuple

import torch

from . import _lib as L


def _shapes(D: int, F: int, E: int, Te: int):
    return [("ln_w", (2, D)), ("ln_b", (2, D)), ("gate_w", (2, E, D)), ("gate_b", (2, E)), ("w1", (2, E, F, D)),
            ("b1", (2, E, F)), ("w2", (2, E, D, F)), ("b2", (2, E, D)), ("st_emb_w", (2 * D, Te)), ("st_emb_b", (2 * D,)),
            ("st_norm_w", (D,)), ("st_norm_b", (D,)), ("st_out_w", (D, D)), ("st_out_b", (D,))]


def reference_keys(prefix: str, E: int) -> Dict[str, list]:
    """tensor name -> the reference state_dict keys stacked into it (branch-major, then expert)."""
    br = lambda fmt: [prefix + fmt.format(b=b) for b in range(2)]
    ex = lambda fmt: [prefix + fmt.format(b=b, e=e) for b in range(2) for e in range(E)]
    return {
        "ln_w": br(".branches.{b}.layernorm.weight"), "ln_b": br(".branches.{b}.layernorm.bias"),
        "gate_w": br(".branches.{b}.moe.gate.weight"), "gate_b": br(".branches.{b}.moe.gate.bias"),
        "w1": ex(".branches.{b}.moe.experts.{e}.0.weight"), "b1": ex(".branches.{b}.moe.experts.{e}.0.bias"),
        "w2": ex(".branches.{b}.moe.experts.{e}.2.weight"), "b2": ex(".branches.{b}.moe.experts.{e}.2.bias"),
        "st_emb_w": [prefix + ".proj_out.emb_layers.1.weight"], "st_emb_b": [prefix + ".proj_out.emb_layers.1.bias"],
        "st_norm_w": [prefix + ".proj_out.norm.weight"], "st_norm_b": [prefix + ".proj_out.norm.bias"],
        "st_out_w": [prefix + ".proj_out.out_layers.2.weight"], "st_out_b": [prefix + ".proj_out.out_layers.2.bias"],
    }


class _Flat:
    """One flat fp32 device buffer with named views (256-byte aligned starts) and the matching MdmMoeTensors struct."""

    def __init__(self, shapes, device):
        offs, n = {}, 0
        for name, shp in shapes:
            offs[name] = n
            n += (math.prod(shp) + 63) // 64 * 64
        self.flat = torch.zeros(n, dtype=torch.float32, device=device)
        self.views = {name: self.flat[offs[name]:offs[name] + math.prod(shp)].view(shp) for name, shp in shapes}
        self.struct = L.MoeTensors(**{name: self.views[name].data_ptr() for name in L.MoeTensors.NAMES})


class MoEFFNTrainer:
    """One ``MoEMultiBranchFFN`` (latent D, expert hidden F, E experts, time-embedding width Te) in training mode."""

    def __init__(self, D: int, F: int, E: int, Te: int, device="cuda", lr: float = 2e-4, betas: Tuple[float, float] = (0.9, 0.999),
                 eps: float = 1e-8, max_norm: float = 1.0, dropout: float = 0.0, seed: int = 0, moe_coef: float = 0.01):
        self.D, self.F, self.E, self.Te = D, F, E, Te
        self.dropout, self.seed = float(dropout), int(seed)  # the mask of iteration i uses seed + i (see forward)
        # coefficient of get_total_moe_loss (transformer.py:265-270), which the reference's training path never calls: the
        # trainer adds and logs the UNSCALED get_moe_loss (gaussian_diffusion.py:985, ddpm_trainer.py:217-222).  Only the extra
        # "loss_moe_scaled" entry of the logged dict uses it.
        self.moe_coef = float(moe_coef)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise L.MdmError("the MoE training step runs on the HIP path only (no CPU fallback)")
        L.lib()  # fail loudly when the extension is missing
        shapes = _shapes(D, F, E, Te)
        self.params, self.grads = _Flat(shapes, self.device), _Flat(shapes, self.device)
        self.adam_m = torch.zeros_like(self.params.flat)
        self.adam_v = torch.zeros_like(self.params.flat)
        self.lr, self.betas, self.eps, self.max_norm = lr, betas, eps, max_norm
        self.step_count = 0
        self._sumsq = torch.zeros(1, dtype=torch.float32, device=self.device)
        self.lb_loss = torch.zeros(2, dtype=torch.float32, device=self.device)
        self._ws, self._shape, self._saved = None, None, None

    # ---- parameters ---------------------------------------------------------------------------------------------------------
    def load_state_dict(self, sd: Dict[str, torch.Tensor], prefix: str):
        """Copy the block's parameters out of a reference state_dict (keys under `prefix`, e.g.
        'decoder_blocks_low.0.module.ffn')."""
        for name, keys in reference_keys(prefix, self.E).items():
            src = torch.stack([sd[k].detach().float() for k in keys]) if len(keys) > 1 else sd[keys[0]].detach().float()
            self.params.views[name].copy_(src.reshape(self.params.views[name].shape))

    def state_dict(self, prefix: str) -> Dict[str, torch.Tensor]:
        out = {}
        for name, keys in reference_keys(prefix, self.E).items():
            v = self.params.views[name]
            lead = 0 if len(keys) == 1 else (1 if len(keys) == 2 else 2)  # stacked: nothing / branches / branches x experts
            rows = v.reshape(len(keys), *v.shape[lead:])
            for k, t in zip(keys, rows):
                out[k] = t.detach().clone()
        return out

    def _same_device(self, *tensors):
        dev = self.params.flat.device
        for t in tensors:
            if t is not None and t.device != dev:
                raise L.MdmError(f"tensor on {t.device}, block on {dev}: the HIP path does not copy across devices")

    # ---- forward / backward ---------------------------------------------------------------------------------------------------
    def _workspace(self, B: int, S: int) -> torch.Tensor:
        if self._shape != (B, S):
            n = L.lib().mdm_moe_train_workspace_bytes(B, S, self.D, self.F, self.E, self.Te)
            if n < 0:
                raise L.MdmError("unsupported MoE training shape")
            self._ws = torch.empty(n, dtype=torch.uint8, device=self.device)
            self._shape = (B, S)
        return self._ws

    def forward(self, x: torch.Tensor, emb: torch.Tensor, eph: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
                route_out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x (B, S, D), emb (B, De); eph = (weight (Te, De), bias (Te)) of the captured per-call projection when De != Te."""
        L.require_cuda(x, emb)
        self._same_device(x, emb, *(eph or ()))
        B, S, D = x.shape
        De = emb.shape[-1]
        if D != self.D or emb.shape[0] != B:
            raise L.MdmError("MoE training forward: shape mismatch")
        if De != self.Te and eph is None:
            raise L.MdmError("emb width differs from the block's time_embed_dim: pass the captured projection (eph)")
        x = x.contiguous().float()
        emb = emb.contiguous().float()
        ew = eb = None
        if De != self.Te:
            ew, eb = eph[0].contiguous().float(), eph[1].contiguous().float()
            if tuple(ew.shape) != (self.Te, De) or tuple(eb.shape) != (self.Te,):
                raise L.MdmError("eph projection has the wrong shape")
            L.require_cuda(ew, eb)
        ws = self._workspace(B, S)
        out = torch.empty_like(x)
        # one mask per optimizer iteration AND per data-parallel rank: the masks are keyed on (seed, site, LOCAL row, element),
        # so ranks that shard a batch must not share a seed or every shard would drop the same pattern
        rank = torch.distributed.get_rank() if torch.distributed.is_available() and torch.distributed.is_initialized() else 0
        mask_seed = (self.seed + self.step_count + rank * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        with torch.cuda.device(self.device):
            L.check(L.lib().mdm_moe_ffn_train_forward(
                C.byref(self.params.struct), self.D, self.F, self.E, self.Te, De, C.c_void_p(L.ptr(ew)), C.c_void_p(L.ptr(eb)),
                C.c_void_p(x.data_ptr()), C.c_void_p(emb.data_ptr()), B, S, C.c_float(self.dropout), C.c_uint64(mask_seed),
                C.c_void_p(out.data_ptr()),
                C.c_void_p(self.lb_loss.data_ptr()), C.c_void_p(L.ptr(route_out)), C.c_void_p(ws.data_ptr()), C.c_int64(ws.numel()),
                C.c_void_p(L.stream_ptr())), "mdm_moe_ffn_train_forward")
        self._saved = (x, emb, ew, B, S, De, mask_seed)
        return out

    def backward(self, dout: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """dL/dout (B, S, D) -> (dL/dx, dL/demb); parameter gradients land in self.grads (overwritten)."""
        if self._saved is None:
            raise L.MdmError("backward() without a forward()")
        x, emb, ew, B, S, De, mask_seed = self._saved
        L.require_cuda(dout)
        self._same_device(dout)
        dout = dout.contiguous().float()
        if tuple(dout.shape) != (B, S, self.D):
            raise L.MdmError("dout has the wrong shape")
        dx, demb = torch.empty_like(x), torch.empty_like(emb)
        with torch.cuda.device(self.device):
            L.check(L.lib().mdm_moe_ffn_train_backward(
                C.byref(self.params.struct), self.D, self.F, self.E, self.Te, De, C.c_void_p(L.ptr(ew)), C.c_void_p(x.data_ptr()),
                C.c_void_p(emb.data_ptr()), B, S, C.c_float(self.dropout), C.c_uint64(mask_seed), C.c_void_p(dout.data_ptr()),
                C.c_void_p(dx.data_ptr()),
                C.c_void_p(demb.data_ptr()), C.byref(self.grads.struct), C.c_void_p(self._ws.data_ptr()),
                C.c_int64(self._ws.numel()), C.c_void_p(L.stream_ptr())), "mdm_moe_ffn_train_backward")
        self._saved = None
        return dx, demb

    # ---- optimizer side (ddpm_trainer.py:228-244) -----------------------------------------------------------------------------
    def all_reduce_grads(self, group=None):
        """Average the gradients over the data-parallel ranks: ONE collective over the flat buffer."""
        all_reduce_mean_(self.grads.flat, group)

    def optimizer_step(self):
        """clip_grad_norm_(max_norm) + Adam, both on the flat buffers, no host sync."""
        self.step_count += 1
        lib, n = L.lib(), self.params.flat.numel()
        with torch.cuda.device(self.device):
            sp = L.stream_ptr()
            L.check(lib.mdm_sumsq(C.c_void_p(self.grads.flat.data_ptr()), C.c_int64(n), C.c_void_p(self._sumsq.data_ptr()),
                                  C.c_void_p(sp)), "mdm_sumsq")
            L.check(lib.mdm_adam_step(C.c_void_p(self.params.flat.data_ptr()), C.c_void_p(self.grads.flat.data_ptr()),
                                      C.c_void_p(self.adam_m.data_ptr()), C.c_void_p(self.adam_v.data_ptr()), C.c_int64(n),
                                      C.c_float(self.lr), C.c_float(self.betas[0]), C.c_float(self.betas[1]), C.c_float(self.eps),
                                      C.c_int32(self.step_count), C.c_void_p(self._sumsq.data_ptr()), C.c_float(self.max_norm),
                                      C.c_void_p(sp)), "mdm_adam_step")

    def grad_norm(self) -> float:
        return float(self._sumsq.sqrt().item())

    def train_step(self, x, emb, target, eph=None, mask=None, group=None) -> Dict[str, float]:
        """One iteration on this block: forward, masked MSE against `target` as backward_G forms it
        (ddpm_trainer.py:201-226: per-frame mean over features, frames weighted by the length mask), backward, gradient
        all-reduce, clip + Adam.  Returns the logged scalars."""
        out = self.forward(x, emb, eph)
        diff = out - target
        B, S, D = out.shape
        if mask is None:
            mask = torch.ones(B, S, device=out.device)
        denom = mask.sum().clamp_min(1.0)
        loss = ((diff * diff).mean(dim=-1) * mask).sum() / denom
        dout = diff * (2.0 / D) * (mask / denom)[..., None]
        self.backward(dout)
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            self.all_reduce_grads(group)
        self.optimizer_step()
        return loss_logs(loss, self.lb_loss.sum(), self.moe_coef)


def loss_logs(loss_mot_rec, moe_loss, moe_coef: float = 0.01) -> Dict[str, float]:
    """The logged scalars of one iteration, keyed and combined as ``DDPMTrainer.backward_G`` does (ddpm_trainer.py:201-226):
    ``loss_moe`` is the UNSCALED load-balancing sum over the block's SwitchMoELayers -- what ``training_losses`` stores
    (gaussian_diffusion.py:985: ``terms['moe_loss'] = model.get_moe_loss(model)``) from THIS forward's routing -- and
    ``loss_total = loss_mot_rec + loss_moe``, the quantity the reference back-propagates (:217, :236).  The 0.01-scaled figure
    of ``get_total_moe_loss`` (transformer.py:265-270, not on the reference's training path) is reported beside them."""
    rec, moe = float(loss_mot_rec), float(moe_loss)
    return {"loss_mot_rec": rec, "loss_moe": moe, "loss_total": rec + moe, "loss_moe_scaled": moe_coef * moe}


def all_reduce_mean_(flat: torch.Tensor, group=None) -> torch.Tensor:
    """In-place mean of one flat gradient buffer over the ranks of `group` (what DDP's bucketed all-reduce computes,
    tools/train.py:140-145).  No-op outside a process group."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return flat
    world = dist.get_world_size(group)
    if world > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat.div_(world)
    return flat
