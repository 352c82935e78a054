"""Synthetic problem batches from a counter-based SplitMix64 stream.

Draw j (1..3) of problem i is mix(seed + 3 i + j): any shard of any batch can be generated
independently, on any rank, and the C oracle generates the same numbers (oracle/ip_oracle.c,
orc_gen_problems) so tests can cross-check the two.
"""
import numpy as np

DIST_MONOTONE, DIST_REFERENCE_LIKE, DIST_NON_MONOTONE = 0, 1, 2

_M1 = np.uint64(0x9E3779B97F4A7C15)
_M2 = np.uint64(0xBF58476D1CE4E5B9)
_M3 = np.uint64(0x94D049BB133111EB)


def _splitmix64(z):
    with np.errstate(over="ignore"):
        z = z + _M1
        z = (z ^ (z >> np.uint64(30))) * _M2
        z = (z ^ (z >> np.uint64(27))) * _M3
        return z ^ (z >> np.uint64(31))


def _u01(seed, ctr):
    with np.errstate(over="ignore"):
        z = _splitmix64(np.uint64(seed) + ctr)
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def generate(seed, first, n, dist=DIST_MONOTONE):
    """Return (pos0, pos1, pos2) float64 arrays for problems first .. first+n-1."""
    idx = np.arange(first, first + n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        base = np.uint64(3) * idx
    u1, u2, u3 = (_u01(seed, base + np.uint64(j)) for j in (1, 2, 3))
    if dist == DIST_MONOTONE:
        pos0 = 1000.0 * u1
        pos1 = pos0 + 10.0 + 500.0 * u2
        pos2 = pos1 + 10.0 + 500.0 * u3
    elif dist == DIST_REFERENCE_LIKE:
        pos0 = np.zeros(n)
        pos1 = 20.0 + 360.0 * u1
        pos2 = np.full(n, 400.0)
    elif dist == DIST_NON_MONOTONE:
        pos0 = -500.0 + 1000.0 * u1
        pos1 = -500.0 + 1000.0 * u2
        pos2 = -500.0 + 1000.0 * u3
    else:
        raise ValueError("unknown distribution %r" % (dist,))
    return pos0, pos1, pos2


def shard_range(n_total, rank, world_size):
    """Contiguous shard [first, first+count) of rank: sizes differ by at most 1."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError("bad rank/world_size")
    q, r = divmod(n_total, world_size)
    first = rank * q + min(rank, r)
    return first, q + (1 if rank < r else 0)
