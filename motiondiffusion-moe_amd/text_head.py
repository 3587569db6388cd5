"""Text side of the path: the reference's ``EnhancedTextEncoder`` (text_encoder.py:6-43) with
its projection head on the HIP library and the DeBERTa backbone supplied by the caller.

The reference constructs the backbone with ``AutoModel.from_pretrained("microsoft/deberta-v3-large")`` -- a network
fetch.  Here the backbone is any ``callable``/``nn.Module`` returning ``last_hidden_state`` and can be loaded from a LOCAL
directory (``EnhancedTextEncoder.from_local``); the learned parts that live in the reference's checkpoints keep their
names (``proj.0.*`` LayerNorm, ``proj.1.*`` Linear, ``prompt_tokens``), so ``text_encoder.*`` keys load unchanged.
Projection head arithmetic (LayerNorm -> Linear -> GELU, prompt concatenation, mean pooling) runs in
``mdm_text_head_forward``; there is no eager fallback."""
from __future__ import annotations

import ctypes as C
from typing import List, Optional

import torch
import torch.nn as nn

from . import _lib as L
from .ops import PackedWeight


class EnhancedTextEncoder(nn.Module):
    num_prompt_tokens = 8  # text_encoder.py:19

    def __init__(self, output_dim: int, dropout: float = 0.1, *, bert=None, tokenizer=None,
                 hidden_size: Optional[int] = None, precision: int = 3, max_length: int = 77):
        super().__init__()
        if hidden_size is None:
            if bert is None or not hasattr(bert, "config"):
                raise ValueError("hidden_size is required when the backbone has no .config.hidden_size")
            hidden_size = int(bert.config.hidden_size)
        if hidden_size > 1024:
            raise L.MdmError("the HIP projection head covers hidden sizes up to 1024 (deberta-v3-large)")
        self.hidden_size, self.output_dim, self.precision, self.max_length = hidden_size, output_dim, precision, max_length
        if isinstance(bert, nn.Module):
            self.bert = bert  # registered: `bert.*` keys of a reference checkpoint load into it
        else:
            self._bert_fn = bert
        self.tokenizer = tokenizer
        # parameter containers with the reference's names (never called: the arithmetic is in the HIP library)
        self.proj = nn.Sequential(nn.LayerNorm(hidden_size), nn.Linear(hidden_size, output_dim), nn.Dropout(dropout),
                                  nn.GELU())
        self.prompt_tokens = nn.Parameter(torch.randn(1, self.num_prompt_tokens, hidden_size))
        self._packed = None
        self.register_load_state_dict_post_hook(lambda mod, _: setattr(mod, "_packed", None))

    @classmethod
    def from_local(cls, path: str, output_dim: int, dropout: float = 0.1, **kw):
        """Backbone + tokenizer from a local directory (no network): what a user with the DeBERTa weights calls."""
        from transformers import AutoModel, AutoTokenizer
        bert = AutoModel.from_pretrained(path, local_files_only=True)
        tok = AutoTokenizer.from_pretrained(path, local_files_only=True)
        return cls(output_dim, dropout, bert=bert, tokenizer=tok, **kw)

    def _apply(self, fn, *a, **k):
        self._packed = None
        return super()._apply(fn, *a, **k)

    def _weights(self):
        w = self.proj[1].weight
        key = (w.data_ptr(), w._version, str(w.device))
        if self._packed is None or self._packed[0] != key:
            self._packed = (key, PackedWeight(w.detach(), with_lo=True))
        return self._packed[1]

    @torch.no_grad()
    def project(self, hidden_states: torch.Tensor):
        """(B, N0, Hs) last_hidden_state -> (pooled (B, Dt), projected (B, 8 + N0, Dt))   (text_encoder.py:31-43)."""
        L.require_cuda(hidden_states)
        if hidden_states.device != self.prompt_tokens.device:
            raise L.MdmError("hidden states and the projection head must live on the same GPU")
        h = hidden_states.detach().to(torch.float32).contiguous()
        B, N0, Hs = h.shape
        assert Hs == self.hidden_size, (Hs, self.hidden_size)
        P, Dt = self.num_prompt_tokens, self.output_dim
        pw = self._weights()
        lib = L.lib()
        nbytes = lib.mdm_text_head_workspace_bytes(B, N0, P, Hs, Dt)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=h.device)
        xf_out = torch.empty(B, P + N0, Dt, device=h.device)
        xf_proj = torch.empty(B, Dt, device=h.device)
        packed = L.Packed(pw.hi.data_ptr(), pw.lo.data_ptr(), pw.Kp)
        prompts = self.prompt_tokens.detach().to(torch.float32).reshape(P, Hs).contiguous()
        ln, lin = self.proj[0], self.proj[1]
        with torch.cuda.device(h.device):
            L.check(lib.mdm_text_head_forward(
                C.c_void_p(h.data_ptr()), C.c_void_p(prompts.data_ptr()), C.c_void_p(ln.weight.data_ptr()),
                C.c_void_p(ln.bias.data_ptr()), C.byref(packed), C.c_void_p(lin.bias.data_ptr()), C.c_int32(B), C.c_int32(N0),
                C.c_int32(P), C.c_int32(Hs), C.c_int32(Dt), C.c_void_p(xf_out.data_ptr()), C.c_void_p(xf_proj.data_ptr()),
                C.c_void_p(ws.data_ptr()), C.c_int64(nbytes), C.c_int32(self.precision), C.c_void_p(L.stream_ptr())),
                "mdm_text_head_forward")
        return xf_proj, xf_out

    @torch.no_grad()
    def forward(self, text: List[str], device):
        bert = getattr(self, "bert", None) or getattr(self, "_bert_fn", None)
        if bert is None or self.tokenizer is None:
            raise L.MdmError("no text backbone attached: the reference fetches microsoft/deberta-v3-large by name "
                             "(text_encoder.py:9-11); load it from a local directory with "
                             "EnhancedTextEncoder.from_local(path, output_dim) or pass bert= / tokenizer=")
        inputs = self.tokenizer(text, padding=True, truncation=True, max_length=self.max_length,
                                return_tensors="pt").to(device)                      # text_encoder.py:25-28
        outputs = bert(input_ids=inputs.input_ids, attention_mask=inputs.attention_mask, return_dict=True)
        return self.project(outputs.last_hidden_state)                               # :38-43
