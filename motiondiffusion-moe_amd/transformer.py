"""Drop-in ``MotionTransformer`` whose forward runs on hand-written gfx950 kernels.

Mirrors the reference module's interface (text2motion/models/transformer.py:166-361): same constructor
keywords, same ``state_dict`` keys/shapes (layout.py), same ``forward(x, timesteps, length, text=None,
xf_proj=None, xf_out=None)``, ``encode_text``, ``generate_src_mask``, ``num_frames``, MoE counter helpers.

What differs, on purpose:
  * the reference builds 33 randomly initialised ``nn.Linear`` layers *inside every forward*
    (stylization.py:22-24, transformer.py:313-315) and draws its Performer feature matrices lazily without saving
    them (fast_attention.py:33-36).  Here both are explicit state: drawn once (``ephemeral_mode="frozen"``, with
    the same generators in the same order as the reference's first forward), injectable
    (``set_ephemerals`` / ``set_projections``) and persistable; ``ephemeral_mode="resample"`` redraws the Linears
    before every forward exactly like the reference does.
  * everything that depends only on the text (cross-attention key/value projections and the linear-attention
    ``softmax(K)^T V`` state) is computed once per distinct ``xf_out`` and cached.
There is no eager / CPU fallback: without the HIP library or a GPU tensor, forward raises.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from . import _lib as L
from .layout import BUFFER_LEAVES, resolve_dims, state_dict_layout
from .packing import PackedModel, format_class
from .synth import ephemeral_names, projection_names


class _Node(nn.Module):
    """Anonymous container; only exists so parameter paths spell the reference's state_dict keys."""


class MotionTransformer(nn.Module):
    def __init__(self, input_feats: int, num_frames: int = 60, latent_dim: int = 512, ff_size: int = 1024,
                 num_layers: int = 4, num_heads: int = 4, dropout: float = 0.1, text_latent_dim: int = 256,
                 moe_num_experts: int = 4, model_size: str = "small", chunk_size: int = 256, text_encoder=None,
                 precision: int = 3, ephemeral_mode: str = "frozen", **kwargs):
        super().__init__()
        D, F_, Dt = resolve_dims(latent_dim, ff_size, text_latent_dim, model_size)
        if D % num_heads:
            raise ValueError("latent_dim must be divisible by num_heads")
        self.input_feats, self.num_frames = input_feats, num_frames
        self.latent_dim, self.ff_size, self.text_latent_dim = D, F_, Dt
        self.num_layers, self.num_heads, self.dropout = num_layers, num_heads, dropout
        self.moe_num_experts, self.chunk_size = moe_num_experts, chunk_size
        self.time_embed_dim = 4 * D
        if precision not in L.PRECISIONS:
            raise ValueError(f"precision must be one of {L.PRECISIONS} (include/mdm_hip.h: MDM_PREC_*)")
        self.precision = precision
        if ephemeral_mode not in ("frozen", "resample"):
            raise ValueError("ephemeral_mode must be 'frozen' or 'resample'")
        self.ephemeral_mode = ephemeral_mode
        self._layout = state_dict_layout(input_feats, num_frames, latent_dim, ff_size, num_layers, num_heads,
                                         text_latent_dim, moe_num_experts, model_size)
        for key, shape in self._layout:
            node = self
            parts = key.split(".")
            for p in parts[:-1]:
                if p not in node._modules:
                    node.add_module(p, _Node())
                node = node._modules[p]
            if parts[-1] in BUFFER_LEAVES:
                node.register_buffer(parts[-1], torch.zeros(shape))
            else:
                node.register_parameter(parts[-1], nn.Parameter(torch.empty(shape)))
        # callable(text, device) -> (xf_proj (B,Dt), xf_out (B,N,Dt)); an nn.Module (text_head.EnhancedTextEncoder) is
        # registered so a reference checkpoint's `text_encoder.*` keys load into it
        if isinstance(text_encoder, nn.Module):
            self.text_encoder = text_encoder
        self.text_encoder_fn = text_encoder
        self._eph: Optional[Dict[str, Tuple[torch.Tensor, torch.Tensor]]] = None
        self._proj: Optional[Dict[str, torch.Tensor]] = None
        self._packed: Dict[str, PackedModel] = {}  # per weight-format class (packing.format_class)
        self._text_cache = None
        self._ws: Optional[torch.Tensor] = None
        self._uncond: Optional[Tuple[torch.Tensor, torch.Tensor]] = None
        self.reset_parameters()
        self.register_load_state_dict_post_hook(lambda mod, _: mod.invalidate())

    # ------------------------------------------------------------------------------------------------
    # parameters / captured randomness
    # ------------------------------------------------------------------------------------------------
    def kernel_cfg(self) -> dict:
        return dict(latent_dim=self.latent_dim, ff_size=self.ff_size, text_latent_dim=self.text_latent_dim,
                    num_heads=self.num_heads, num_layers=self.num_layers, moe_num_experts=self.moe_num_experts,
                    input_feats=self.input_feats, num_frames=self.num_frames)

    @torch.no_grad()
    def reset_parameters(self):
        """Same distributions as the reference constructors: nn.Linear/Conv defaults, LayerNorm (1,0), randn
        sequence embedding, xavier_normal(gain=0.1) on every >=2-D Performer parameter (fast_attention.py:132-135),
        zeros for the MoE gates, cross-attn gates, StylizationBlock out layers and the output head
        (switch_moe.py:28-29, fast_attention.py:240,267, stylization.py:17, transformer.py:257)."""
        lay = dict(self._layout)
        for key, p in self.state_dict(keep_vars=True).items():
            leaf = key.rsplit(".", 1)[-1]
            performer = ".local_attn." in key or ".global_attn." in key
            if leaf in BUFFER_LEAVES:
                p.zero_()
            elif key == "sequence_embedding":
                p.normal_()
            elif leaf in ("gate", "adaptive_gate") or ".moe.gate." in key or key.startswith("out."):
                p.zero_()
            elif ".out_layers.2." in key and not performer:
                p.zero_()
            elif p.dim() == 1:
                wshape = lay.get(key[:-len(leaf)] + "weight")
                if leaf == "weight":
                    p.fill_(1.0)  # LayerNorm gain
                elif wshape is not None and len(wshape) == 1:
                    p.zero_()  # LayerNorm bias
                else:  # Linear / Conv bias: U(-1/sqrt(fan_in), 1/sqrt(fan_in))
                    fan_in = wshape[1] * (wshape[2] if len(wshape) == 3 else 1)
                    p.uniform_(-1 / math.sqrt(fan_in), 1 / math.sqrt(fan_in))
            elif performer:
                nn.init.xavier_normal_(p, gain=0.1)
            else:  # kaiming_uniform(a=sqrt(5)) == U(-1/sqrt(fan_in), 1/sqrt(fan_in))
                fan_in = p.shape[1] * (p.shape[2] if p.dim() == 3 else 1)
                p.uniform_(-1 / math.sqrt(fan_in), 1 / math.sqrt(fan_in))
        self.invalidate()

    def invalidate(self):
        """Forget packed weights / caches (call after mutating parameters in place)."""
        self._packed = {}
        self._text_cache = None
        self._time_table = None
        if not getattr(self, "_uncond_explicit", False):
            self._uncond = None  # encoded by the (possibly reloaded) text encoder: must be re-encoded

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self.invalidate()
        return r

    @property
    def device(self) -> torch.device:
        return self.sequence_embedding.device

    def set_ephemerals(self, items):
        """items: ordered iterable of (name, weight, bias) or {name: (weight, bias)} for synth.ephemeral_names()."""
        d = {n: (w, b) for n, w, b in items} if not isinstance(items, dict) else dict(items)
        need = ephemeral_names(self.num_layers, self.text_latent_dim != self.latent_dim)
        missing = [n for n in need if n not in d]
        if missing:
            raise ValueError(f"missing ephemeral projections: {missing[:3]}...")
        self._eph = {n: d[n] for n in need}
        self.invalidate()

    def set_projections(self, items):
        d = dict(items)
        need = projection_names(self.num_layers)
        dh = self.latent_dim // self.num_heads
        for n in need:
            if tuple(d[n].shape) != (dh, min(dh, 256)):
                raise ValueError(f"projection {n} must be ({dh},{min(dh, 256)})")
        self._proj = {n: d[n] for n in need}
        self.invalidate()

    def captured_state(self) -> dict:
        """The random state the reference never saves; store it next to a checkpoint to reproduce samples."""
        return {"ephemerals": self._eph, "projections": self._proj}

    @torch.no_grad()
    def draw_ephemerals(self):
        """Draw the per-forward Linears the way the reference does: nn.Linear(...) constructed on the CPU,
        consuming the CPU default generator, in call order (stylization.py:23, transformer.py:314)."""
        eph = {}
        for n in ephemeral_names(self.num_layers, self.text_latent_dim != self.latent_dim):
            lin = nn.Linear(self.text_latent_dim, self.latent_dim) if n == "text_proj" else \
                nn.Linear(self.latent_dim, self.time_embed_dim)
            eph[n] = (lin.weight.detach().clone(), lin.bias.detach().clone())
        self._eph = eph
        self.invalidate()

    @torch.no_grad()
    def draw_projections(self):
        """fast_attention.py:19-27: QR of randn(dh, 256) drawn on the module's device, column-normalised, * dh^-1/4."""
        dh = self.latent_dim // self.num_heads
        proj = {}
        for n in projection_names(self.num_layers):
            g = torch.randn(dh, 256, device=self.device)
            q, _ = torch.linalg.qr(g, mode="reduced")
            proj[n] = torch.nn.functional.normalize(q, dim=0) * (dh ** -0.25)
        self._proj = proj
        self.invalidate()

    # ------------------------------------------------------------------------------------------------
    # reference helper API
    # ------------------------------------------------------------------------------------------------
    def moe_buffers(self) -> Dict[str, torch.Tensor]:
        return {k: v for k, v in self.named_buffers() if k.rsplit(".", 1)[-1] in BUFFER_LEAVES}

    def reset_all_moe_counters(self, model=None):
        for b in (model or self).moe_buffers().values():
            b.zero_()

    @staticmethod
    def load_balancing_loss(usage: torch.Tensor, importance: torch.Tensor, epsilon: float = 1e-8) -> torch.Tensor:
        """SwitchMoELayer.get_load_balancing_loss (switch_moe.py:113-145): E * (1 - sum_e usage_frac[e] * importance_frac[e])."""
        u = usage / usage.sum().clamp_min(epsilon)
        i = importance / importance.sum().clamp_min(epsilon)
        return usage.numel() * (1.0 - (u * i).sum())

    def get_moe_loss(self, model=None):
        """Sum of the load-balancing losses of every SwitchMoE layer (transformer.py:272-279), from the device-side counters
        the router kernels maintain; a value without a graph, like the reference's (its counters are updated under
        no_grad, switch_moe.py:71-92)."""
        m = model or self
        bufs = m.moe_buffers()
        total = torch.zeros((), device=self.device)
        for k, usage in bufs.items():
            if not k.endswith("expert_usage"):
                continue
            total = total + self.load_balancing_loss(usage, bufs[k.replace("expert_usage", "expert_importance")])
        return total

    def get_total_moe_loss(self, model=None, moe_coef=0.01):
        return moe_coef * self.get_moe_loss(model)

    def encode_text(self, text: List[str], device):
        if self.text_encoder_fn is None:
            raise L.MdmError(
                "no text encoder attached: the reference's DeBERTa-v3-large weights are a network fetch "
                "(text_encoder.py:9-11) and are out of scope here; pass text_encoder=callable(text, device) -> "
                "(xf_proj, xf_out), or call forward with xf_proj/xf_out")
        return self.text_encoder_fn(text, device)

    def set_uncond_embedding(self, xf_proj: torch.Tensor, xf_out: torch.Tensor):
        """Cached embedding of the empty caption used by classifier-free guidance (one row, broadcast over B)."""
        self._uncond = (xf_proj, xf_out)
        self._uncond_explicit = True

    def uncond_embedding(self, B: int, device):
        if self._uncond is None:
            xp, xo = self.encode_text([""], device)
            self._uncond = (xp[:1].contiguous(), xo[:1].contiguous())
        xp, xo = self._uncond
        return xp.to(device).expand(B, -1).contiguous(), xo.to(device).expand(B, -1, -1).contiguous()

    def generate_src_mask(self, T: int, length: torch.Tensor) -> torch.Tensor:
        return (torch.arange(T, device=length.device)[None, :] < length[:, None]).to(torch.float32)

    # ------------------------------------------------------------------------------------------------
    # HIP path
    # ------------------------------------------------------------------------------------------------
    def pack(self) -> PackedModel:
        if self.device.type != "cuda":
            raise L.MdmError("MotionTransformer runs on hand-written HIP kernels only: move it to a GPU "
                             "(no CPU/eager fallback exists)")
        if self._eph is None:
            self.draw_ephemerals()
        if self._proj is None:
            self.draw_projections()
        cls = format_class(self.precision)
        if cls not in self._packed:
            sd = {k: v.detach() for k, v in self.state_dict().items()}
            self._packed[cls] = PackedModel(sd, self.kernel_cfg(), self._eph, self._proj, self.device,
                                            with_lo=True, counters=self.moe_buffers(), precision=self.precision)
            self._text_cache = None
        return self._packed[cls]

    def workspace_bytes(self, B: int, T: int, N: int) -> int:
        pm = self.pack()
        need = L.lib().mdm_workspace_bytes(C.byref(pm.model), C.c_int32(B), C.c_int32(T), C.c_int32(N))
        if need < 0:
            raise L.MdmError("unsupported model shape for the HIP path")
        return int(need)

    def new_workspace(self, B: int, T: int, N: int) -> torch.Tensor:
        """A private scratch buffer (concurrent forwards on different streams must not share one)."""
        return torch.empty(self.workspace_bytes(B, T, N), dtype=torch.uint8, device=self.device)

    def _workspace(self, B: int, T: int, N: int) -> torch.Tensor:
        need = self.workspace_bytes(B, T, N)
        if self._ws is None or self._ws.numel() < need or self._ws.device != self.device:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    def prepare_text(self, xf_out: torch.Tensor, private: bool = False, ntok=None):
        """Build (or fetch) the text-side cache for this xf_out [B,N,Dt].  ``private=True`` returns a cache object
        owned by the caller (pass it back as ``forward(..., text_cache=)``) instead of the module's single slot.
        ``ntok`` (optional, one int per sample, 1 <= ntok[b] <= N): sample b's own token count; its rows past that are
        padding that neither cross-attention sees (include/mdm_hip.h: MdmTextCache.ntok)."""
        pm = self.pack()
        xf_out = xf_out.detach().to(device=self.device, dtype=torch.float32).contiguous()
        B, N, Dt = xf_out.shape
        if Dt != self.text_latent_dim:
            raise ValueError(f"xf_out last dim {Dt} != text_latent_dim {self.text_latent_dim}")
        nt_key = None
        if ntok is not None:
            nt_host = [int(v) for v in (ntok.tolist() if torch.is_tensor(ntok) else ntok)]
            if len(nt_host) != B or min(nt_host) < 1 or max(nt_host) > N:
                raise ValueError("ntok must hold one token count in [1, N] per sample")
            nt_key = tuple(nt_host)
            if all(v == N for v in nt_host):
                ntok, nt_key = None, None
        key = (xf_out.data_ptr(), xf_out._version, B, N, self.precision, nt_key)
        if not private and self._text_cache is not None and self._text_cache["key"] == key:
            return self._text_cache
        D, H, L2 = self.latent_dim, self.num_heads, 2 * self.num_layers
        dh = D // H
        dev = self.device
        at = torch.empty((L2, B, H, dh, dh), dtype=torch.float32, device=dev)
        sk = torch.empty((L2, B, N, D), dtype=torch.float32, device=dev)
        sv = torch.empty((L2, B, N, D), dtype=torch.float32, device=dev)
        tc = L.TextCache()
        tc.lin_at, tc.sd_k, tc.sd_v, tc.B, tc.N = at.data_ptr(), sk.data_ptr(), sv.data_ptr(), B, N
        nt_dev = None
        if nt_key is not None:
            nt_dev = torch.tensor(nt_key, dtype=torch.int32, device=dev)
            tc.ntok = nt_dev.data_ptr()
        fold = ()
        npass = L.lib().mdm_sd_fold_passes(D, H, N) if self.precision in (L.PREC_BF16, L.PREC_F16, L.PREC_FP8) else 0
        if npass > 0:
            # throughput modes: query / output projections of the text cross-attention folded into the text side, in passes
            # of <= 128 folded columns (one pass up to N = 32 text tokens, four at the reference's 85)
            h16 = torch.bfloat16 if self.precision == L.PREC_BF16 else torch.float16
            fold = (torch.zeros((L2, B, npass, 128, D), dtype=h16, device=dev),
                    torch.zeros((L2, B, npass, 128), dtype=torch.float32, device=dev),
                    torch.zeros((L2, B, npass, D, 128), dtype=h16, device=dev))
            tc.sd_kfold, tc.sd_cb, tc.sd_vfold = (t.data_ptr() for t in fold)
        ws = self._workspace(B, 2, N)
        with torch.cuda.device(dev):  # launches go to the current stream OF THE MODEL'S DEVICE, whatever device is current
            L.check(L.lib().mdm_text_cache_build(C.byref(pm.model), C.c_void_p(xf_out.data_ptr()), C.byref(tc),
                                                 C.c_void_p(ws.data_ptr()), C.c_int64(ws.numel()), C.c_int32(self.precision),
                                                 C.c_void_p(L.stream_ptr())), "mdm_text_cache_build")
        cache = {"key": key, "tc": tc, "keep": (at, sk, sv, xf_out, nt_dev) + fold, "B": B, "N": N, "pm": pm}
        if not private:
            self._text_cache = cache
        return cache

    def stem_cache(self, steps: int, xf_proj: torch.Tensor):
        """Per-loop stem cache (include/mdm_hip.h: MdmStemCache): the time-embedding chain tabulated for every integer
        timestep of a ``steps``-long schedule (cached per module) + the text half of the gated fusion for this batch."""
        pm = self.pack()
        dev, D = self.device, self.latent_dim
        key = (steps, id(pm))
        if getattr(self, "_time_table", None) is None or self._time_table[0] != key:
            self._time_table = (key, torch.empty((steps, D), dtype=torch.float32, device=dev))
            fill_table = True
        else:
            fill_table = False
        table = self._time_table[1]
        xp = xf_proj.detach().to(device=dev, dtype=torch.float32).contiguous()
        gx = torch.empty((xp.shape[0], D), dtype=torch.float32, device=dev)
        ws = self._workspace(128, 2, 1)
        with torch.cuda.device(dev):
            L.check(L.lib().mdm_stem_cache_build(C.byref(pm.model), C.c_int32(steps), C.c_void_p(table.data_ptr() if fill_table else 0),
                                                 C.c_void_p(xp.data_ptr()), C.c_int32(xp.shape[0]), C.c_void_p(gx.data_ptr()),
                                                 C.c_void_p(ws.data_ptr()), C.c_int64(ws.numel()), C.c_int32(self.precision),
                                                 C.c_void_p(L.stream_ptr())), "mdm_stem_cache_build")
        sc = L.StemCache()
        sc.time_table, sc.gx, sc.steps = table.data_ptr(), gx.data_ptr(), steps
        return {"sc": sc, "keep": (table, gx)}

    @torch.no_grad()
    def forward(self, x: torch.Tensor, timesteps: torch.Tensor, length: torch.Tensor,
                text: Optional[List[str]] = None, xf_proj=None, xf_out=None, *, forced_routing=None,
                trace: bool = False, out: Optional[torch.Tensor] = None, stem_cache=None, text_cache=None,
                workspace: Optional[torch.Tensor] = None, text_tokens=None):
        if not x.is_cuda:
            raise L.MdmError("MotionTransformer.forward needs GPU tensors: the denoiser runs on HIP kernels only")
        if x.device != self.device:
            raise L.MdmError(f"input on {x.device} but the model (packed weights, workspace) lives on {self.device}")
        B, T, Fe = x.shape
        if Fe != self.input_feats:
            raise ValueError(f"expected {self.input_feats} features, got {Fe}")
        if T % 2:
            raise ValueError("T must be even: the reference's 2-scale U-shape cannot add skip and upsampled "
                             "tensors of different lengths (transformer.py:223-224,353)")
        if T > self.num_frames:
            raise ValueError(f"T={T} exceeds num_frames={self.num_frames}")
        if xf_proj is None or xf_out is None:
            xf_proj, xf_out = self.encode_text(text, x.device)
        if self.ephemeral_mode == "resample":
            self.draw_ephemerals()
        pm = self.pack()
        tcache = (text_cache if text_cache is not None and text_cache.get("pm") is pm
                  else self.prepare_text(xf_out, ntok=text_tokens))
        if tcache["B"] != B:
            raise ValueError("xf_out batch does not match x")
        dev = x.device
        x = x.detach().to(torch.float32).contiguous()
        ts = timesteps.detach().to(device=dev, dtype=torch.int64).contiguous()
        ln = length.detach().to(device=dev, dtype=torch.int32).contiguous()
        xp = xf_proj.detach().to(device=dev, dtype=torch.float32).contiguous()
        ws = workspace if workspace is not None else self._workspace(B, T, tcache["N"])
        if ws.numel() < self.workspace_bytes(B, T, tcache["N"]):
            raise ValueError("workspace too small for this (B, T, N)")
        if out is None:
            out = torch.empty((B, T, Fe), dtype=torch.float32, device=dev)
        fr = None
        if forced_routing is not None:
            fr = forced_routing.to(device=dev, dtype=torch.int32).contiguous()
            if fr.numel() and (int(fr.min()) < 0 or int(fr.max()) >= self.moe_num_experts):  # test hook, host-checked
                raise ValueError("forced_routing holds expert indices outside [0, moe_num_experts)")
        tr = torch.zeros((2 * self.num_layers, 4, B * T, self.latent_dim), dtype=torch.float32, device=dev) if trace else None
        with torch.cuda.device(dev):
            L.check(L.lib().mdm_denoiser_forward(
                C.byref(pm.model), C.byref(tcache["tc"]), C.c_void_p(x.data_ptr()), C.c_void_p(ts.data_ptr()),
                C.c_void_p(ln.data_ptr()), C.c_void_p(xp.data_ptr()), C.c_int32(B), C.c_int32(T), C.c_void_p(out.data_ptr()),
                C.c_void_p(ws.data_ptr()), C.c_int64(ws.numel()), C.c_void_p(L.ptr(fr)), C.c_void_p(L.ptr(tr)),
                C.byref(stem_cache["sc"]) if stem_cache is not None else None,
                C.c_int32(self.precision), C.c_void_p(L.stream_ptr())), "mdm_denoiser_forward")
        if trace:
            return out, tr
        return out
