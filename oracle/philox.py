"""TEST INFRASTRUCTURE ONLY (the checker, never the product path).

Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11; the Random123
library's philox4x32_R with R = 10) in numpy, and the integer draws the HIP kernels derive from it.  Pinned by the
known-answer vectors of Random123's kat_vectors file (tests/test_oracle_golden.py).

The reference draws its training timesteps with torch.randint (base_trainer.py:202): the build draws them in-kernel
from this counter-based generator instead, so the oracle restates the generator and the mapping to 1..T."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Counters: uint32 arrays (broadcastable); key: two python ints.  Returns four uint32 arrays."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) & MASK for c in (c0, c1, c2, c3))
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2
        n0 = (p1 >> np.uint64(32)) ^ c1 ^ np.uint64(k0)
        n1 = p1 & MASK
        n2 = (p0 >> np.uint64(32)) ^ c3 ^ np.uint64(k1)
        n3 = p0 & MASK
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return tuple(c.astype(np.uint32) for c in (c0, c1, c2, c3))


def sample_timesteps(n: int, T: int, seed: int, stream_id: int) -> np.ndarray:
    """t[i] in 1..T: word i % 4 of Philox block i // 4 of the (seed, stream) sequence, scaled by multiply-high."""
    seed &= 0xFFFFFFFFFFFFFFFF
    q = np.arange((n + 3) // 4, dtype=np.uint64)
    words = np.stack(philox4x32_10(q, 0, stream_id, 0x3C6EF372, seed & 0xFFFFFFFF, seed >> 32), axis=1).reshape(-1)[:n]
    return (1 + ((words.astype(np.uint64) * np.uint64(T)) >> np.uint64(32))).astype(np.int64)


def normal(n: int, seed: int, stream_id: int, first_index: int = 0) -> np.ndarray:
    """N(0,1) samples of GLOBAL flat elements first_index .. first_index + n - 1 (both multiples of 4) of the (seed,
    stream) sequence: Philox block q = element // 4, words (x, y) and (z, w) through Box-Muller with u in (0, 1] for the
    radius and [0, 1) for the angle, outputs ordered (r0 cos, r0 sin, r1 cos, r1 sin).  float32 arithmetic with numpy's
    log/sin/cos: the HIP kernel's fast log/sincos differ from these in the last bits, so compare with a tolerance."""
    assert n % 4 == 0 and first_index % 4 == 0
    seed &= 0xFFFFFFFFFFFFFFFF
    q = np.arange(n // 4, dtype=np.uint64) + np.uint64(first_index // 4)
    x, y, z, w = philox4x32_10(q & MASK, q >> np.uint64(32), stream_id, 0x6A09E667, seed & 0xFFFFFFFF, seed >> 32)
    k = np.float32(2.3283064365386963e-10)
    f = lambda v: v.astype(np.float32)          # noqa: E731  (uint32 -> float32 rounds to nearest, as the device cast does)
    u0, u1, u2, u3 = (f(x) + np.float32(1)) * k, f(y) * k, (f(z) + np.float32(1)) * k, f(w) * k
    ra, rb = np.sqrt(np.float32(-2) * np.log(u0)), np.sqrt(np.float32(-2) * np.log(u2))
    two_pi = np.float32(6.283185307179586)
    out = np.stack([ra * np.cos(two_pi * u1), ra * np.sin(two_pi * u1), rb * np.cos(two_pi * u3), rb * np.sin(two_pi * u3)], axis=1)
    return out.reshape(-1).astype(np.float32)
