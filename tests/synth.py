"""Deterministic synthetic embeddings for tests and golden fixtures.

Integer-only generator (splitmix64 counter hash -> sum of four 16-bit uniforms, Irwin-Hall ~ normal),
so the same (n, d, seed) gives bit-identical arrays on any machine / numpy version: the golden
fixtures store only outputs plus a sha256 of the regenerated inputs.
"""
import hashlib

import numpy as np

_M = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M
        return z ^ (z >> np.uint64(31))


def normal_f32(n: int, d: int, seed: int, scale: float) -> np.ndarray:
    """(n, d) float32, approximately N(0, scale^2), exactly reproducible."""
    out = np.empty((n, d), dtype=np.float32)
    step = max(1, (1 << 22) // d)
    with np.errstate(over="ignore"):
        base = np.uint64(seed) * np.uint64(0xD1342543DE82EF95)
    for r0 in range(0, n, step):
        r1 = min(n, r0 + step)
        with np.errstate(over="ignore"):
            idx = (np.arange(r0 * d, r1 * d, dtype=np.uint64) + base) & _M
        z = _splitmix64(idx)
        u = ((z & np.uint64(0xFFFF)) + ((z >> np.uint64(16)) & np.uint64(0xFFFF))
             + ((z >> np.uint64(32)) & np.uint64(0xFFFF)) + ((z >> np.uint64(48)) & np.uint64(0xFFFF))).astype(np.int64)
        x = (u - 2 * 65535).astype(np.float64) * (1.7320508075688772 / 65535.0)   # unit variance
        out[r0:r1] = (x * scale).astype(np.float32).reshape(r1 - r0, d)
    return out


def passages_f16(n: int, d: int = 768, seed: int = 1) -> np.ndarray:
    """approximately unit-norm rows (iid N(0, 1/d)), fp16"""
    return normal_f32(n, d, seed, 1.0 / np.sqrt(d)).astype(np.float16)


def queries_f32(b: int, d: int = 768, seed: int = 2) -> np.ndarray:
    return normal_f32(b, d, seed, 1.0)


def sha(*arrays) -> str:
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()
