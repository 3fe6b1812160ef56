"""numpy restatement of the device dropout RNG (csrc/common.h: tell_hash32 /
tell_keep) so that tests can rebuild the exact keep-masks a kernel used."""
import numpy as np

M32 = np.uint64(0xFFFFFFFF)


def hash32(seed, salt, idx):
    idx = np.asarray(idx, dtype=np.uint64)
    lo = idx & M32
    hi = idx >> np.uint64(32)
    x = (lo * np.uint64(0x9E3779B1) + np.uint64(seed)) & M32
    y = (hi * np.uint64(0x85EBCA77) + np.uint64(salt) * np.uint64(0xC2B2AE3D) + np.uint64(0x27D4EB2F)) & M32

    def mix(v):
        v = v ^ (v >> np.uint64(16))
        v = (v * np.uint64(0x7FEB352D)) & M32
        v = v ^ (v >> np.uint64(15))
        v = (v * np.uint64(0x846CA68B)) & M32
        v = v ^ (v >> np.uint64(16))
        return v
    x = mix(x ^ y)
    x = (x + y) & M32
    return mix(x)


def threshold(p):
    return np.uint64(min(max(int(float(np.float32(p)) * 4294967296.0), 0), 4294967295))


def keep_mask(seed, salt, n_or_idx, p):
    """float32 array of 0/1 keep flags for element indices 0..n-1 (or the given indices)."""
    idx = np.arange(n_or_idx, dtype=np.uint64) if np.isscalar(n_or_idx) else n_or_idx
    return (hash32(seed, salt, idx) >= threshold(p)).astype(np.float32)
