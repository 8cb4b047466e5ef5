"""TEST INFRASTRUCTURE (only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import oracle/): CPU restatement of
the library's in-kernel draws (uwudiff_amd/csrc/objective.hip: philox4x32_10, philox_u01, philox_normal4, schedule_draw_kernel).

The reference draws with torch (diffusion.py:68-70 ``torch.randint``, :75 ``torch.randn_like``; rectified_flow.py:37
``torch.rand``); its CPU (mt19937) and GPU (cuRAND-style Philox) streams already differ from each other, so a parity run injects the
draws (RNG note).  What is pinned here is the GENERATOR: Philox4x32-10 as published (Salmon, Moraes, Dror,
Shaw, "Parallel random numbers: as easy as 1, 2, 3", SC'11), checked against the known-answer vectors of the Random123 distribution
(tests/test_oracle_loss.py), and the library's mapping of its words to t / u01 / N(0, 1)."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10_words(c, k):
    """``c``: uint32 [n, 4] counters, ``k``: uint32 [n, 2] keys -> uint32 [n, 4] (ten rounds, the key bumped between rounds)."""
    c = np.array(c, dtype=np.uint32).reshape(-1, 4).copy()
    k = np.array(k, dtype=np.uint32).reshape(-1, 2).copy()
    for _ in range(10):
        p0 = M0 * c[:, 0].astype(np.uint64)
        p1 = M1 * c[:, 2].astype(np.uint64)
        hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), (p0 & MASK).astype(np.uint32)
        hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), (p1 & MASK).astype(np.uint32)
        c = np.stack([hi1 ^ c[:, 1] ^ k[:, 0], lo1, hi0 ^ c[:, 3] ^ k[:, 1], lo0], axis=1)
        k = np.stack([k[:, 0] + W0, k[:, 1] + W1], axis=1)
    return c


def philox(n_counters, seed, offset):
    """The library's stream: counter i = (offset + i) as the low 64 bits (words 0, 1), words 2, 3 zero; key = seed."""
    ctr = np.uint64(offset) + np.arange(n_counters, dtype=np.uint64)
    c = np.stack([(ctr & MASK).astype(np.uint32), (ctr >> np.uint64(32)).astype(np.uint32),
                  np.zeros(n_counters, np.uint32), np.zeros(n_counters, np.uint32)], axis=1)
    k = np.tile(np.array([[seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF]], dtype=np.uint32), (n_counters, 1))
    return philox4x32_10_words(c, k)


def u01(words):
    return ((words >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -24)


def normal(n, seed, offset):
    """n (multiple of 4) N(0, 1) values: Box-Muller on word pairs (0, 1) and (2, 3) of each counter (fp32 like the kernel)."""
    w = philox(n // 4, seed, offset)
    u = u01(w).astype(np.float64)
    rad0, rad1 = np.sqrt(-2.0 * np.log(u[:, 0])), np.sqrt(-2.0 * np.log(u[:, 2]))
    a0, a1 = 2.0 * np.pi * u[:, 1], 2.0 * np.pi * u[:, 3]
    z = np.stack([rad0 * np.cos(a0), rad0 * np.sin(a0), rad1 * np.cos(a1), rad1 * np.sin(a1)], axis=1)
    return z.reshape(-1).astype(np.float32)


def timesteps(B, n_train, seed, offset):
    w = philox((B + 3) // 4, seed, offset).reshape(-1)[:B]
    return ((w.astype(np.uint64) * np.uint64(n_train)) >> np.uint64(32)).astype(np.int64)


def uniform(B, seed, offset):
    return u01(philox((B + 3) // 4, seed, offset).reshape(-1)[:B])
