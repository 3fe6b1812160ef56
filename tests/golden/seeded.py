"""Deterministic large tensors for the golden fixtures (test infrastructure).

Tensors above `THRESH` elements are not stored by value in the `.npz` files:
`make_golden.py` overwrites them (before running the reference) with
`scale * randn(shape, generator=manual_seed(crc32(name)))` and stores only
`(scale, shape)`; the tests rebuild the identical values with `regen`.
torch's CPU generator is bit-reproducible for a given torch build (the GPU box
runs the same image).
"""
import zlib

import numpy as np
import torch

THRESH = 1024
SUB_N = 4096


def subsample(a):
    """Strided subsample (<= SUB_N+1 values) of a large expected output; tests apply
    the same function to their own result before comparing."""
    flat = a.reshape(-1)
    if flat.shape[0] <= SUB_N:
        return flat
    return flat[::flat.shape[0] // SUB_N]


def _gen(name):
    return torch.Generator().manual_seed(zlib.crc32(name.encode()) & 0x7FFFFFFF)


def make(name, shape, scale, positive=False):
    t = torch.randn(*shape, generator=_gen(name)) * float(scale)
    return t.abs() if positive else t


def fill_(tensor, name, positive=False):
    """Overwrite `tensor` in place; returns the fixture record (scale, shape, positive)."""
    scale = float(np.float32(max(float(tensor.float().std()), 1e-3)))
    tensor.copy_(make(name, tuple(tensor.shape), scale, positive))
    return np.array([scale, float(positive)] + list(tensor.shape), dtype=np.float64)


def regen(name, record):
    scale, positive = float(record[0]), bool(record[1])
    shape = tuple(int(s) for s in record[2:])
    return make(name, shape, np.float32(scale), positive)


def load_npz(path):
    """-> dict group -> {key: torch tensor}; seeded entries are regenerated."""
    z = np.load(path)
    out = {}
    fname = path.split('/')[-1]
    aliases = []
    for k in z.files:
        grp, key = k.split('/', 1)
        if grp.startswith('alias_'):
            aliases.append((grp[len('alias_'):], key, str(z[k])))
            continue
        if grp.startswith('sub_'):
            out.setdefault('sub', {})[key] = torch.from_numpy(z[k])
            continue
        if grp.startswith('seeded_'):
            grp = grp[len('seeded_'):]
            val = regen(fname + ':' + grp + '/' + key, z[k])
        else:
            a = z[k]
            val = torch.from_numpy(a) if a.shape != () else a.item()
        out.setdefault(grp, {})[key] = val
    for grp, key, target in aliases:
        out[grp][key] = out[grp][target]
    return out
