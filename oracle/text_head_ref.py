"""CPU restatement of the reference's text projection head (TEST INFRASTRUCTURE ONLY -- never imported by the product).

Follows text2motion/models/text_encoder.py:
  * proj = LayerNorm(hidden) -> Linear(hidden, output_dim) -> Dropout -> GELU          (:13-18; dropout is a no-op in eval)
  * prompts = prompt_tokens.repeat(B, 1, 1), prepended to the encoder's last_hidden_state (:31-40)
  * projected = proj(cat([prompts, hidden_states], dim=1)); pooled = projected.mean(dim=1)   (:41-43)
Pinned by tests/golden/text_head.npz, produced by the reference's own forward with the DeBERTa weight fetch stubbed
(oracle/make_golden.py::case_text_head).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def text_head(hidden_states: torch.Tensor, prompt_tokens: torch.Tensor, ln_w, ln_b, w, b):
    """hidden_states (B, N0, Hs), prompt_tokens (1, P, Hs) -> (pooled (B, Dt), projected (B, P + N0, Dt))."""
    B = hidden_states.shape[0]
    h = torch.cat([prompt_tokens.repeat(B, 1, 1), hidden_states], dim=1)      # :31,40
    h = F.layer_norm(h, (h.shape[-1],), ln_w, ln_b, 1e-5)                     # proj.0
    projected = F.gelu(F.linear(h, w, b))                                     # proj.1, proj.3 (exact erf GELU)
    return projected.mean(dim=1), projected                                   # :42-43
