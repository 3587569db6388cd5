"""MI355X-native denoising hot path of MotionDiffusion-MoE.

The directory name is not a Python identifier; import it as
    importlib.import_module("motiondiffusion-moe_amd")      or     import mdm_amd   (alias at the repo root)
"""
__all__ = ["synth", "layout"]
