"""CPU emulation: what per-32-element e8m0 (OCP MX) block scales would buy the e4m3 expert MLP over the per-row / per-channel
fp32 scales csrc/gemm8.hip applies today.  Both operands of both GEMMs are quantised; the error is measured against fp64.
Result (printed): the error is set by e4m3's 3 mantissa bits (rms relative rounding error 2^-4 / sqrt(3) per operand, ~5 % of a
random-sign dot product whatever its length), not by the scale granularity -- block scales only help range, and LayerNorm
outputs / GELU hidden units / trained weights do not have a range problem inside a row."""
import torch


def q_row(t):  # one fp32 scale per row: amax -> 448 (what mdm_pack_fp8 and the router's row normalisation do)
    s = t.abs().amax(-1, keepdim=True).clamp_min(1e-30) / 448
    return (t / s).to(torch.float8_e4m3fn).float() * s


def q_mx(t, spec=True):  # one power-of-two scale per 32 elements along K
    tb = t.reshape(*t.shape[:-1], -1, 32)
    am = tb.abs().amax(-1, keepdim=True).clamp_min(1e-30)
    if spec:  # OCP MX v1.0: 2^(floor(log2 amax) - emax_elem), emax(e4m3) = 8; values in (448, 512) saturate
        s = torch.exp2(torch.floor(torch.log2(am)) - 8)
    else:     # smallest power of two that avoids saturation
        s = torch.exp2(torch.ceil(torch.log2(am / 448)))
    return ((tb / s).clamp(-448, 448).to(torch.float8_e4m3fn).float() * s).reshape(t.shape)


def main():
    torch.manual_seed(0)
    M, D, F = 4096, 1024, 2048
    x = torch.nn.functional.layer_norm(torch.randn(M, D) * torch.rand(M, 1) * 3, (D,))
    x[:, :8] *= 12  # a few outlier channels, as LayerNorm outputs of trained transformers have
    w1 = (torch.rand(F, D) * 2 - 1) * D ** -0.5
    w2 = (torch.rand(D, F) * 2 - 1) * F ** -0.5
    ref = torch.nn.functional.gelu(x.double() @ w1.double().T) @ w2.double().T
    for name, q in (("f16", lambda t: t.half().float()), ("e4m3, row / channel fp32 scales", q_row),
                    ("e4m3, MX per-32 e8m0 (spec rounding)", q_mx), ("e4m3, per-32 power of two, no saturation", lambda t: q_mx(t, False))):
        y = (q(torch.nn.functional.gelu(q(x) @ q(w1).T)) @ q(w2).T).double()
        row = (y - ref).abs().amax(-1) / ref.abs().max()
        print(f"{name:42s} rel-inf {float((y - ref).abs().max() / ref.abs().max()):.2e}   median row error {float(row.median()):.2e}   "
              f"rms relative {float((y - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()):.2e}")


if __name__ == "__main__":
    main()
