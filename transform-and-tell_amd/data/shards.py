"""On-disk shard format of the data plane: the MongoDB documents + JPEG directory of the reference's
readers (nytimes_faces_ner_matched.py:81-227) replaced by self-contained `.npz` shards, one file per few thousand
samples, already tokenised (RoBERTa BPE ids) and decoded (uint8 pixels of the 224 x 224 crop that
scripts/process_images.py:37-39 produces).  Ragged fields are stored concatenated with an offset vector.

    <dir>/<split>-00000.npz
        context_ids  int32 [sum L]      context_off  int64 [N+1]     (ids incl. <s> ... </s>, <= 512 each)
        caption_ids  int32 [sum T]      caption_off  int64 [N+1]
        context_copy int8  [sum L]      caption_copy int8 [sum T]    (entity copy masks of the indexer; optional)
        image        uint8 [N,224,224,3]
        face_embeds  float32 [sum F,512]   face_off int64 [N+1]      (F <= 4; 0 rows = the reference's empty [1,0] field)
        n_person_names int32 [N]                                       (PERSON entities in the caption, :125-130; optional)
        obj_embeds   float32 [sum O,2048]  obj_off  int64 [N+1]      (O <= 64; absent when use_objects is false)
        metadata     str [N]                                          (JSON: caption, context, web_url, image_path, ...)

Faces and `use_caption_names` (nytimes_faces_ner_matched.py:125-130,168-170): the reference keeps only as many faces as
the caption has PERSON names.  A shard either records that count per sample (`n_person_names`: the reader trims) or its
writer has ALREADY trimmed `face_embeds` to it - a shard without the field is taken as pre-trimmed, and the reader
checks the only invariant it can (F <= 4, the largest count any config reaches)."""
import glob
import json
import os

import numpy as np


def write_shard(path, samples):
    """samples: list of dicts with keys context_ids, caption_ids, image (uint8 HWC), face_embeds [F,512] (F may be 0),
    optional obj_embeds [O,2048], context_copy / caption_copy, metadata (dict)."""
    def ragged(key, dtype, width=None):
        parts = [np.asarray(s[key], dtype=dtype).reshape(-1, width) if width else np.asarray(s[key], dtype=dtype)
                 for s in samples]
        off = np.zeros(len(samples) + 1, dtype=np.int64)
        off[1:] = np.cumsum([len(p) for p in parts])
        cat = np.concatenate(parts) if parts else np.zeros((0, width) if width else (0,), dtype=dtype)
        return cat, off
    out = {}
    out['context_ids'], out['context_off'] = ragged('context_ids', np.int32)
    out['caption_ids'], out['caption_off'] = ragged('caption_ids', np.int32)
    if all('context_copy' in s for s in samples):
        out['context_copy'], _ = ragged('context_copy', np.int8)
        out['caption_copy'], _ = ragged('caption_copy', np.int8)
    out['image'] = np.stack([np.asarray(s['image'], dtype=np.uint8) for s in samples])
    out['face_embeds'], out['face_off'] = ragged('face_embeds', np.float32, 512)
    if all('n_person_names' in s for s in samples):
        out['n_person_names'] = np.array([int(s['n_person_names']) for s in samples], dtype=np.int32)
    if all(s.get('obj_embeds') is not None for s in samples):
        out['obj_embeds'], out['obj_off'] = ragged('obj_embeds', np.float32, 2048)
    out['metadata'] = np.array([json.dumps(s.get('metadata', {})) for s in samples])
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    np.savez(path, **out)


def shard_paths(directory, split):
    return sorted(glob.glob(os.path.join(directory, '%s-*.npz' % split)))


def _map_file(path):
    """Read-only mapping of a whole file as a uint8 array that pins NO file descriptor: np.memmap / mmap.mmap keep a
    duplicated descriptor for the life of the mapping, so every live shard cost one; libc's mmap keeps the pages after
    close().  The mapping is released when the last view of the returned array dies."""
    import ctypes
    import mmap as _mmap
    import weakref
    size = os.path.getsize(path)
    libc = ctypes.CDLL(None, use_errno=True)
    libc.mmap.restype = ctypes.c_void_p
    libc.mmap.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_long]
    libc.munmap.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    fd = os.open(path, os.O_RDONLY)
    try:
        addr = libc.mmap(None, size, _mmap.PROT_READ, _mmap.MAP_PRIVATE, fd, 0)
    finally:
        os.close(fd)
    if addr in (None, ctypes.c_void_p(-1).value):
        raise OSError(ctypes.get_errno(), 'mmap failed', path)
    buf = (ctypes.c_ubyte * size).from_address(addr)
    arr = np.frombuffer(buf, dtype=np.uint8)
    arr.flags.writeable = False
    weakref.finalize(buf, libc.munmap, addr, size)            # (views keep `buf` alive through arr.base)
    return arr


def _mapped_arrays(path):
    """{member: array} of an UNCOMPRESSED .npz (what np.savez writes) without reading it: a stored zip member is the
    bytes of its .npy file at a fixed offset, so every numeric array becomes a read-only VIEW of ONE mapping of the shard
    file - no copy, no CRC pass (np.load spends 15 ms per 128-sample shard on zlib.crc32 alone; 40 % of the loader thread).
    One mapping per shard that pins no descriptor (_map_file): a np.memmap keeps a duplicated file descriptor until its
    last view dies, and the reader's instances hold views of image / face_embeds / obj_embeds - three descriptors per live
    shard exhausted the default limit of 1024 on a pass over a few hundred shards (EMFILE inside the loader thread).
    -> None when a member is compressed or of a dtype that cannot be mapped (strings are read normally)."""
    import struct
    import zipfile
    out = {}
    whole = None
    with zipfile.ZipFile(path) as zf, open(path, 'rb') as f:
        for info in zf.infolist():
            name = info.filename[:-4] if info.filename.endswith('.npy') else info.filename
            if info.compress_type != zipfile.ZIP_STORED:
                return None
            f.seek(info.header_offset)
            hdr = f.read(30)
            if hdr[:4] != b'PK\x03\x04':
                return None
            n_name, n_extra = struct.unpack('<HH', hdr[26:30])
            f.seek(info.header_offset + 30 + n_name + n_extra)
            version = np.lib.format.read_magic(f)
            shape, fortran, dtype = (np.lib.format.read_array_header_1_0(f) if version == (1, 0)
                                     else np.lib.format.read_array_header_2_0(f))
            if dtype.hasobject or fortran:
                return None
            if dtype.kind in 'US' or 0 in shape:                 # strings (metadata) / empty arrays: a normal read
                with zf.open(info) as m:
                    out[name] = np.lib.format.read_array(m, allow_pickle=False)
                continue
            if whole is None:
                whole = _map_file(path)                                    # one mapping per shard, no descriptor kept
            start, nbytes = f.tell(), int(np.prod(shape)) * dtype.itemsize
            out[name] = np.ndarray(shape, dtype=dtype, buffer=whole, offset=start) if start + nbytes <= whole.size else None
            if out[name] is None:
                return None
    return out


def read_shard(path, mmap=True):
    """-> list of sample dicts (numpy views into the shard's arrays; memory-mapped when the shard is uncompressed)."""
    arrays = _mapped_arrays(path) if mmap else None
    if arrays is None:
        z = np.load(path, allow_pickle=False)
        arrays = {k: z[k] for k in z.files}
    n = len(arrays['context_off']) - 1
    has_obj, has_copy = 'obj_embeds' in arrays, 'context_copy' in arrays
    out = []
    for i in range(n):
        c0, c1 = arrays['context_off'][i:i + 2]
        t0, t1 = arrays['caption_off'][i:i + 2]
        f0, f1 = arrays['face_off'][i:i + 2]
        s = {'context_ids': arrays['context_ids'][c0:c1], 'caption_ids': arrays['caption_ids'][t0:t1],
             'image': arrays['image'][i], 'face_embeds': arrays['face_embeds'][f0:f1],
             'metadata': json.loads(str(arrays['metadata'][i]))}
        if 'n_person_names' in arrays:
            s['n_person_names'] = int(arrays['n_person_names'][i])
        if has_copy:
            s['context_copy'] = arrays['context_copy'][c0:c1]
            s['caption_copy'] = arrays['caption_copy'][t0:t1]
        if has_obj:
            o0, o1 = arrays['obj_off'][i:i + 2]
            s['obj_embeds'] = arrays['obj_embeds'][o0:o1]
        out.append(s)
    return out
