"""TEST INFRASTRUCTURE (oracle): numpy restatement of csrc/noise.hip -- Philox4x32-10 (Salmon et al., SC'11: multipliers
0xD2511F53 / 0xCD9E8D57, Weyl key increments 0x9E3779B9 / 0xBB67AE85) + Box-Muller -- the counter-based generator the HIP
sampler uses where the reference calls th.randn / th.randn_like (gaussian_diffusion.py:1119,1094).  There is no reference
arithmetic to pin here (the reference's noise is whatever the torch global generator yields); the pin is the published
Philox known-answer vector checked in tests/test_host_logic.py.  Only tests/ may import this module."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
STREAM_XT = 0x7FFFFFFF


def philox4x32_10(c, k0, k1):
    """c: uint32 array (..., 4); k0, k1: uint32 scalars -> uint32 (..., 4)."""
    c = np.array(c, dtype=np.uint32, copy=True)
    k0, k1 = np.uint32(k0), np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = M0 * c[..., 0].astype(np.uint64)
            p1 = M1 * c[..., 2].astype(np.uint64)
            n0 = (p1 >> np.uint64(32)).astype(np.uint32) ^ c[..., 1] ^ k0
            n1 = p1.astype(np.uint32)
            n2 = (p0 >> np.uint64(32)).astype(np.uint32) ^ c[..., 3] ^ k1
            n3 = p0.astype(np.uint32)
            c = np.stack([n0, n1, n2, n3], -1)
            k0, k1 = np.uint32(k0 + W0), np.uint32(k1 + W1)
    return c


def uniform_bits(per_sample, nsamples, sample0, seed, stream):
    quads = (per_sample + 3) // 4
    qd = np.arange(quads, dtype=np.uint64)[None, :].repeat(nsamples, 0)
    gs = (np.uint64(sample0) + np.arange(nsamples, dtype=np.uint64))[:, None].repeat(quads, 1)
    c = np.stack([qd.astype(np.uint32), gs.astype(np.uint32), (gs >> np.uint64(32)).astype(np.uint32),
                  np.full_like(qd, stream, dtype=np.uint64).astype(np.uint32)], -1)
    return philox4x32_10(c, np.uint32(seed & 0xFFFFFFFF), np.uint32((seed >> 32) & 0xFFFFFFFF))


def normal(per_sample, nsamples, sample0, seed, stream):
    """float32 (nsamples, per_sample): what mdm_noise_normal writes."""
    b = uniform_bits(per_sample, nsamples, sample0, seed, stream)
    u = ((b >> np.uint32(8)).astype(np.float32) + np.float32(1.0)) * np.float32(1.0 / 16777216.0)
    out = np.empty(b.shape, np.float32)
    for h in range(2):
        u1, u2 = u[..., 2 * h].astype(np.float64), u[..., 2 * h + 1].astype(np.float64)
        r = np.sqrt(-2.0 * np.log(u1))
        a = np.float64(np.float32(6.283185307179586)) * u2
        out[..., 2 * h] = (r * np.cos(a)).astype(np.float32)
        out[..., 2 * h + 1] = (r * np.sin(a)).astype(np.float32)
    return out.reshape(nsamples, -1)[:, :per_sample]
