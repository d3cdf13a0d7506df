"""TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg) — never imported by dmhomo_amd.

CPU restatement (numpy) of dmhomo_amd's sample-indexed noise generator, ``dmh_rng_indexed`` (csrc/rng.hip).

The reference has no counterpart for the VALUES: it draws ``torch.randn(shape)`` (CFG:679), ``torch.randn_like(img)``
(CFG:705) and ``torch.zeros(B).uniform_(0, 1)`` (CFG:90) from the process's stream generator, so its row b depends on how
many rows the process holds.  What is restated here is the published counter-based generator the kernel uses instead —
Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11; the Random123
distribution's known-answer vectors, ``kat_vectors``, are the pin: tests/test_oracle_golden.py) — with
    key     = (seed lo, seed hi)
    counter = (element // 4, draw index, sample id lo, sample id hi)
and, per group of four 32-bit words, two Box-Muller pairs in fp32 (uniforms placed as cuRAND places them,
x * 2^-32 + 2^-33) or the top 24 bits * 2^-24 for a uniform in [0, 1).
"""
import numpy as np

_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = 0x9E3779B9, 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(counter, key):
    """counter (..., 4) uint32, key (2,) ints -> (..., 4) uint32."""
    c = [np.asarray(counter[..., i], dtype=np.uint64) for i in range(4)]
    k0, k1 = int(key[0]) & 0xFFFFFFFF, int(key[1]) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = _M0 * c[0], _M1 * c[2]
        hi0, lo0 = p0 >> np.uint64(32), p0 & _MASK
        hi1, lo1 = p1 >> np.uint64(32), p1 & _MASK
        c = [hi1 ^ c[1] ^ np.uint64(k0), lo1, hi0 ^ c[3] ^ np.uint64(k1), lo0]
        k0, k1 = (k0 + _W0) & 0xFFFFFFFF, (k1 + _W1) & 0xFFFFFFFF
    return np.stack(c, axis=-1).astype(np.uint32)


def words(seed, sample_ids, draw, per_sample):
    """the raw words of one draw: (B, per_sample) uint32 (``dmh_rng_indexed`` kind 2)."""
    sample_ids = np.asarray(list(sample_ids), dtype=np.uint64)
    nq = (per_sample + 3) // 4
    ctr = np.zeros((len(sample_ids), nq, 4), dtype=np.uint32)
    ctr[..., 0] = np.arange(nq, dtype=np.uint32)[None, :]
    ctr[..., 1] = np.uint32(draw & 0xFFFFFFFF)
    ctr[..., 2] = (sample_ids & _MASK).astype(np.uint32)[:, None]
    ctr[..., 3] = (sample_ids >> np.uint64(32)).astype(np.uint32)[:, None]
    w = philox4x32_10(ctr, (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))
    return w.reshape(len(sample_ids), nq * 4)[:, :per_sample]


def _u01(x):
    f = np.float32
    return x.astype(np.float32) * f(2.3283064365386963e-10) + f(1.1641532182693481e-10)


def randn(seed, sample_ids, draw, shape):
    """(B, *shape) fp32 standard normals of draw ``draw`` (kind 0)."""
    per = int(np.prod(shape))
    nq4 = (per + 3) // 4 * 4
    w = words(seed, sample_ids, draw, nq4).reshape(-1, nq4 // 4, 4)
    out = np.empty(w.shape, dtype=np.float32)
    for a, b in ((0, 1), (2, 3)):
        r = np.sqrt(np.float32(-2.0) * np.log(_u01(w[..., a])))
        th = (np.float32(2.0) * _u01(w[..., b])).astype(np.float64) * np.pi      # sincospi(2u): exact argument reduction
        out[..., a] = r * np.cos(th).astype(np.float32)
        out[..., b] = r * np.sin(th).astype(np.float32)
    return out.reshape(len(w), nq4)[:, :per].reshape((len(w),) + tuple(shape))


def uniform(seed, sample_ids, draw, per_sample=1):
    """(B, per_sample) fp32 uniforms in [0, 1) (kind 1)."""
    w = words(seed, sample_ids, draw, per_sample)
    return (w >> np.uint32(8)).astype(np.float32) * np.float32(5.9604644775390625e-8)


# Random123 kat_vectors, philox4x32 with 10 rounds: (counter, key, expected)
KAT = [
    ((0x00000000, 0x00000000, 0x00000000, 0x00000000), (0x00000000, 0x00000000),
     (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff), (0xffffffff, 0xffffffff),
     (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]
