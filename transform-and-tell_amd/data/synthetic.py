"""Synthetic batches with the tensor contract of the reference's readers
(tell/data/dataset_readers/nytimes_faces_ner_matched.py:192-227 +
tell/data/token_indexers/roberta_indexer.py:89-109,185-200):
  context  {'roberta': int64 [B, S]}   <s>=0 first, </s>=2 last real token, right-padded with 1
  image    float32 [B, 3, 224, 224]    ImageNet-normalised pixels (random normal here)
  caption  {'roberta': int64 [B, T+1]}
  face_embeds float32 [B, F<=4, 512]   NaN-padded;  obj_embeds float32 [B, O<=64, 2048] NaN-padded
MongoDB / JPEG / BPE plumbing is out of scope (no data offline)."""
import torch

from ..common.registrable import Registrable


def _ids(g, B, L, lens, vocab, band_mix=None):
    ids = torch.ones(B, L, dtype=torch.long)
    for b in range(B):
        n = int(lens[b])
        if band_mix is None:
            body = torch.randint(4, vocab, (n - 2,), generator=g)
        else:   # 60 % head band, 25 % first tail, 15 % second tail (exercises all clusters)
            u = torch.rand(n - 2, generator=g)
            c0, c1 = band_mix
            body = torch.where(u < 0.6, torch.randint(4, c0, (n - 2,), generator=g),
                               torch.where(u < 0.85, torch.randint(c0, c1, (n - 2,), generator=g),
                                           torch.randint(c1, vocab, (n - 2,), generator=g)))
        ids[b, 0] = 0
        ids[b, 1:n - 1] = body
        ids[b, n - 1] = 2
    return ids


def synthetic_batch(B=16, article_len=512, caption_len=33, faces_objects=False, seed=1234, vocab=50265,
                    variable=False, device='cpu', cutoffs=(5000, 20000)):
    g = torch.Generator().manual_seed(seed)
    if variable:
        alen = torch.randint(min(128, max(article_len // 4, 3)), article_len + 1, (B,), generator=g)
        clen = torch.randint(min(9, max(caption_len // 2, 3)), caption_len + 1, (B,), generator=g)
    else:
        alen = torch.full((B,), article_len)
        clen = torch.full((B,), caption_len)
    batch = {
        'context': {'roberta': _ids(g, B, article_len, alen, vocab).to(device)},
        'image': torch.randn(B, 3, 224, 224, generator=g).to(device),
        'caption': {'roberta': _ids(g, B, caption_len, clen, vocab, cutoffs).to(device)},
    }
    if faces_objects:
        faces = torch.nn.functional.normalize(torch.randn(B, 4, 512, generator=g), dim=-1)
        objs = torch.randn(B, 64, 2048, generator=g).abs()
        nf = torch.randint(0, 5, (B,), generator=g)
        no = torch.randint(0, 65, (B,), generator=g)
        for b in range(B):
            faces[b, int(nf[b]):] = float('nan')
            objs[b, int(no[b]):] = float('nan')
        batch['face_embeds'] = faces.to(device)
        batch['obj_embeds'] = objs.to(device)
    return batch


class DatasetReader(Registrable):
    pass


@DatasetReader.register('synthetic')
class SyntheticReader(DatasetReader):
    """Whole synthetic batches of the input contract (bench.py / tests); the readers under the reference's own
    registration names live in data/readers.py."""

    def __init__(self, use_objects=False, batch_size=16, article_len=512, caption_len=33, seed=1234,
                 device='cuda', **unused):
        self.faces_objects = use_objects
        self.batch_size, self.article_len, self.caption_len = batch_size, article_len, caption_len
        self.seed, self.device = seed, device

    def read(self, n_batches):
        for i in range(n_batches):
            yield synthetic_batch(self.batch_size, self.article_len, self.caption_len, self.faces_objects,
                                  self.seed + i, device=self.device)
