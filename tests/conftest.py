import importlib
import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# a GPU box shows every host core to a process that is given a share of them: torch's default thread count then oversubscribes, and
# the small CPU ops the oracle and the seeded weight builders are made of run several times slower
torch.set_num_threads(min(16, os.cpu_count() or 1))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def pkg(sub=None):
    """The package directory is 'motiondiffusion-moe_amd' (not an identifier) -> importlib."""
    name = "motiondiffusion-moe_amd" + ("." + sub if sub else "")
    return importlib.import_module(name)


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(bytes(z["__meta__"]).decode())
    t = {k: torch.from_numpy(z[k]) for k in z.files if k != "__meta__"}
    return t, meta


def rel_inf(a, b):
    """north_star metric: ||a-b||_inf / ||b||_inf."""
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


_STATE_MEMO = {}


def golden_state(meta):
    """Rebuild (state_dict, eph dict, proj dict, model cfg) for a golden case from its seeds (memoised per case: the tensors are
    shared between tests and treated as read-only; the dicts are fresh, so a test may rebind keys)."""
    key = json.dumps({"cfg": meta["cfg"], "wseed": meta["wseed"], "D": meta["latent_dim"], "Dt": meta["text_latent_dim"]}, sort_keys=True)
    if key not in _STATE_MEMO:
        _STATE_MEMO[key] = _golden_state(meta)
    sd, eph, proj, mcfg = _STATE_MEMO[key]
    return dict(sd), dict(eph), dict(proj), dict(mcfg)


def _golden_state(meta):
    synth = pkg("synth")
    model = pkg("layout")
    cfg = meta["cfg"]
    keys = model.state_dict_layout(
        input_feats=cfg["input_feats"], num_frames=cfg["num_frames"], latent_dim=cfg["latent_dim_arg"],
        ff_size=cfg["ff_size_arg"], num_layers=cfg["num_layers"], num_heads=cfg["num_heads"],
        text_latent_dim=cfg["text_latent_dim_arg"], moe_num_experts=cfg["moe_num_experts"],
        model_size=cfg["model_size"])
    sd = synth.synth_state_dict(keys, meta["wseed"])
    D, Dt = meta["latent_dim"], meta["text_latent_dim"]
    eph = {n: (w, b) for n, w, b in synth.synth_ephemerals(D, Dt, cfg["num_layers"], meta["wseed"])}
    proj = dict(synth.synth_projections(D // cfg["num_heads"], cfg["num_layers"], meta["wseed"]))
    mcfg = dict(latent_dim=D, num_heads=cfg["num_heads"], num_layers=cfg["num_layers"],
                moe_num_experts=cfg["moe_num_experts"])
    return sd, eph, proj, mcfg


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def build_module(meta, device="cuda", precision=3):
    """Our MotionTransformer for a golden case: synthetic weights + injected captured randomness."""
    T = pkg("transformer")
    cfg = meta["cfg"]
    sd, eph, proj, mcfg = golden_state(meta)
    m = T.MotionTransformer(cfg["input_feats"], num_frames=cfg["num_frames"], latent_dim=cfg["latent_dim_arg"],
                            ff_size=cfg["ff_size_arg"], num_layers=cfg["num_layers"], num_heads=cfg["num_heads"],
                            text_latent_dim=cfg["text_latent_dim_arg"], moe_num_experts=cfg["moe_num_experts"],
                            model_size=cfg["model_size"], precision=precision)
    m.load_state_dict(sd, strict=True)
    m.set_ephemerals(eph)
    m.set_projections(proj)
    return m.to(device).eval(), (sd, eph, proj, mcfg)
