"""numpy restatement of the device dropout RNG (csrc/common.h: tell_hash32 /
tell_keep: one hash per aligned index pair, 16-bit halves) so that tests can rebuild the exact keep-masks a kernel used."""
import numpy as np

M32 = np.uint64(0xFFFFFFFF)


def hash32(seed, salt, idx):
    idx = np.asarray(idx, dtype=np.uint64)
    lo = idx & M32
    hi = idx >> np.uint64(32)
    x = (lo * np.uint64(0x9E3779B1) + np.uint64(seed)) & M32
    y = (hi * np.uint64(0x85EBCA77) + np.uint64(salt) * np.uint64(0xC2B2AE3D) + np.uint64(0x27D4EB2F)) & M32
    v = x ^ y
    v = v ^ (v >> np.uint64(16))
    v = (v * np.uint64(0x7FEB352D)) & M32
    v = v ^ (v >> np.uint64(15))
    v = (v * np.uint64(0x846CA68B)) & M32
    v = v ^ (v >> np.uint64(16))
    return v


def threshold(p):
    """16-bit drop threshold (0 = no dropout)."""
    return np.uint64(min(max(int(float(np.float32(p)) * 65536.0), 0), 65535))


def keep_mask(seed, salt, n_or_idx, p):
    """float32 array of 0/1 keep flags for element indices 0..n-1 (or the given indices): element idx uses
    the 16-bit half (idx & 1) of hash32(idx >> 1)."""
    idx = np.arange(n_or_idx, dtype=np.uint64) if np.isscalar(n_or_idx) else np.asarray(n_or_idx, dtype=np.uint64)
    h = hash32(seed, salt, idx >> np.uint64(1))
    bits = np.where((idx & np.uint64(1)) == 1, h >> np.uint64(16), h & np.uint64(0xFFFF))
    return (bits >= threshold(p)).astype(np.float32)
