"""Deterministic synthetic inputs shared by make_golden.py and the tests.

Pure integer arithmetic (a 64-bit mix hash of the element index), so the same values come out
under any numpy/torch version; values lie on a 2^-12 lattice in [-1, 1) and are exact in fp16/fp32.
Large fixture inputs are regenerated from (shape, seed) instead of being stored.
"""
import numpy as np


def hash_u32(n, seed):
    off = (int(seed) * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    with np.errstate(over="ignore"):
        i = np.arange(n, dtype=np.uint64) + np.uint64(off)
        i ^= i >> np.uint64(30)
        i *= np.uint64(0xBF58476D1CE4E5B9)
        i ^= i >> np.uint64(27)
        i *= np.uint64(0x94D049BB133111EB)
        i ^= i >> np.uint64(31)
    return (i >> np.uint64(32)).astype(np.uint32)


def lattice_uniform(shape, seed):
    """float32 array, values k/4096 with k in [-4096, 4096)."""
    n = int(np.prod(shape))
    u = hash_u32(n, seed) >> np.uint32(19)  # 13 bits
    return ((u.astype(np.int32) - 4096).astype(np.float32) / np.float32(4096.0)).reshape(shape)


def lattice_normalish(shape, seed):
    """Sum of four lattice uniforms (bell-shaped, std ~1.15), still exact in fp32 (2^-12 lattice)."""
    return (lattice_uniform(shape, seed) + lattice_uniform(shape, seed + 1000003)
            + lattice_uniform(shape, seed + 2000003) + lattice_uniform(shape, seed + 3000017))


def homography_flow(B, G, seed, jitter=0.002, scale=1.0):
    """Smooth homography-like normalised flow (B,2,G,G) float32 + small lattice jitter."""
    lin = (np.arange(G, dtype=np.float64) * 2 + 1) / G - 1
    gy, gx = np.meshgrid(lin, lin, indexing="ij")
    out = np.empty((B, 2, G, G), np.float32)
    for b in range(B):
        a = 0.9 + 0.04 * b
        den = 1.0 + 0.05 * gx - 0.03 * gy
        fx = (a * gx + 0.08 * gy + 0.03) / den
        fy = (-0.05 * gx + (1.05 - 0.03 * b) * gy - 0.02) / den
        out[b, 0] = (scale * fx).astype(np.float32)
        out[b, 1] = (scale * fy).astype(np.float32)
    return out + np.float32(jitter) * lattice_uniform((B, 2, G, G), seed)
