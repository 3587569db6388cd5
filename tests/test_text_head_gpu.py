"""GPU: the text projection head (mdm_text_head_forward) against the golden produced by the reference's own
EnhancedTextEncoder.forward (text_encoder.py:24-43) with the DeBERTa fetch stubbed (oracle/make_golden.py)."""
import importlib
import os
import sys
import types

import pytest
import torch

from conftest import load_golden, pkg, rel_inf

pytestmark = pytest.mark.gpu


def _head(g, meta, **kw):
    TH = pkg("text_head")
    enc = TH.EnhancedTextEncoder(meta["Dt"], hidden_size=meta["Hs"], **kw)
    enc.load_state_dict({k[3:]: v for k, v in g.items() if k.startswith("sd/")}, strict=True)
    return enc.cuda()


@pytest.mark.parametrize("precision,tol", [(3, 1e-4), (1, 2e-2)])
def test_projection_head_matches_reference_golden(precision, tol):
    g, meta = load_golden("text_head")
    enc = _head(g, meta, precision=precision)
    pooled, projected = enc.project(g["hidden"].cuda())
    assert projected.shape == g["projected"].shape and pooled.shape == g["pooled"].shape
    assert rel_inf(projected.cpu(), g["projected"]) < tol
    assert rel_inf(pooled.cpu(), g["pooled"]) < tol


def test_oracle_agrees_and_forward_uses_backbone():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    R = importlib.import_module("oracle.text_head_ref")
    g, meta = load_golden("text_head")
    B, N0, Hs = g["hidden"].shape

    class Tok:
        def __call__(self, text, **kw):
            assert kw["max_length"] == 77 and kw["padding"] and kw["truncation"]
            ids = torch.zeros(len(text), N0, dtype=torch.long)
            ns = types.SimpleNamespace(input_ids=ids, attention_mask=torch.ones_like(ids))
            ns.to = lambda device: ns
            return ns

    hidden = g["hidden"].cuda()
    bert = lambda input_ids, attention_mask, return_dict: types.SimpleNamespace(last_hidden_state=hidden)
    enc = _head(g, meta, bert=bert, tokenizer=Tok())
    pooled, projected = enc(["a person walks"] * B, torch.device("cuda"))
    sd = {k[3:]: v for k, v in g.items() if k.startswith("sd/")}
    p_ref, o_ref = R.text_head(g["hidden"], sd["prompt_tokens"], sd["proj.0.weight"], sd["proj.0.bias"],
                               sd["proj.1.weight"], sd["proj.1.bias"])
    assert rel_inf(projected.cpu(), o_ref) < 1e-4 and rel_inf(pooled.cpu(), p_ref) < 1e-4
    # a wide backbone (deberta-v3-large width) and a ragged batch
    TH = pkg("text_head")
    big = TH.EnhancedTextEncoder(256, hidden_size=1024).cuda()
    h = torch.randn(5, 21, 1024, device="cuda")
    pooled, projected = big.project(h)
    p_ref, o_ref = R.text_head(h.cpu(), big.prompt_tokens.detach().cpu(), big.proj[0].weight.detach().cpu(),
                               big.proj[0].bias.detach().cpu(), big.proj[1].weight.detach().cpu(),
                               big.proj[1].bias.detach().cpu())
    assert rel_inf(projected.cpu(), o_ref) < 1e-4 and rel_inf(pooled.cpu(), p_ref) < 1e-4


def test_missing_backbone_and_cpu_are_refused():
    L, TH = pkg("_lib"), pkg("text_head")
    enc = TH.EnhancedTextEncoder(32, hidden_size=64)
    with pytest.raises(L.MdmError):
        enc(["x"], torch.device("cuda"))
    with pytest.raises(L.MdmError):
        enc.project(torch.zeros(1, 2, 64))
