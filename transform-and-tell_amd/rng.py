"""numpy restatement of the device dropout RNG (csrc/common.h: tell_quad_* / tell_keep_field: one hash per aligned
index quad, four 16-bit fields) so that tests can rebuild the exact keep-masks a kernel used."""
import numpy as np

M32 = np.uint64(0xFFFFFFFF)
_U = np.uint64


def keep_field(seed, salt, idx):
    """The 16-bit field element idx is judged by (tell_keep_field)."""
    idx = np.asarray(idx, dtype=np.uint64)
    quad = idx >> _U(2)
    lo = quad & M32
    hi = quad >> _U(32)
    x = (lo * _U(0x9E3779B1) + _U(seed) * _U(0x85EBCA6B)) & M32
    y = (hi * _U(0x85EBCA77) + _U(salt) * _U(0xC2B2AE3D) + _U(0x27D4EB2F)) & M32
    h = x ^ y                                                # tell_quad_mix
    h = h ^ (h >> _U(16))
    h = (h * _U(0x7FEB352D)) & M32
    h = h ^ (h >> _U(15))
    a = ((h & _U(0xFFFFFF)) * _U(0xD1B54B)) & M32            # tell_quad_a: elements 0, 1
    a = a ^ (a >> _U(15))
    b = ((h >> _U(8)) * _U(0xA54FF5)) & M32                  # tell_quad_b: elements 2, 3
    b = b ^ (b >> _U(15))
    w = np.where((idx & _U(2)) == 0, a, b)
    return np.where((idx & _U(1)) == 1, w >> _U(16), w & _U(0xFFFF))


def threshold(p):
    """16-bit drop threshold (0 = no dropout)."""
    return np.uint64(min(max(int(float(np.float32(p)) * 65536.0), 0), 65535))


def keep_mask(seed, salt, n_or_idx, p):
    """float32 array of 0/1 keep flags for element indices 0..n-1 (or the given indices)."""
    idx = np.arange(n_or_idx, dtype=np.uint64) if np.isscalar(n_or_idx) else np.asarray(n_or_idx, dtype=np.uint64)
    return (keep_field(seed, salt, idx) >= threshold(p)).astype(np.float32)
