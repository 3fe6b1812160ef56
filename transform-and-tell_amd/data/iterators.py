"""The `bucket` iterator of the configs (config.yaml:99-110; AllenNLP 0.9 BucketIterator, third-party, restated from its
documented behaviour) and the collation of instances into the batch tensors of the model's input contract
(SURVEY 8-a15): ids padded right with 1, copy masks with -1, face / object arrays NaN-padded to the longest of the
batch (the ArrayFields' padding_value=np.nan, nytimes_faces_ner_matched.py:213,217), images normalised on the GPU."""
import random

import numpy as np
import torch

from ..common.registrable import Registrable

MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)        # nytimes_faces_ner_matched.py:67-69


class DataIterator(Registrable):
    pass


def padding_length(instance, field, key='num_tokens'):
    f = instance[field]
    if isinstance(f, dict):
        return len(next(iter(f.values())))
    return int(np.asarray(f).shape[0])


def normalize_images(images_u8, device):
    """uint8 [B,H,W,3] -> float32 [B,3,H,W]: ToTensor + Normalize on the device (one HIP kernel on the GPU)."""
    x = images_u8.to(device) if torch.is_tensor(images_u8) else torch.as_tensor(np.ascontiguousarray(images_u8)).to(device)
    B, H, W, C = x.shape
    if x.is_cuda:
        from .. import hip
        out = torch.empty(B, C, H, W, dtype=torch.float32, device=device)
        hip.call('tell_image_normalize', x, out, B, H, W, MEAN[0], MEAN[1], MEAN[2], STD[0], STD[1], STD[2])
        return out
    x = x.permute(0, 3, 1, 2).float() / 255.0
    return (x - torch.tensor(MEAN).view(1, 3, 1, 1)) / torch.tensor(STD).view(1, 3, 1, 1)


def collate_host(instances, padding_value=1, pin=False):
    """list of reader instances -> the batch as HOST tensors (ids padded, faces / objects NaN-padded, pixels still uint8
    NHWC): everything of `collate` that needs no device, so that a loader thread can run it one batch ahead.
    pin=True: page-locked buffers (the copies of `to_device` are then asynchronous)."""
    def host(arr):
        t = torch.from_numpy(arr)
        return t.pin_memory() if pin and torch.cuda.is_available() else t
    batch = {}
    for field in ('context', 'caption'):
        keys = instances[0][field].keys()
        n = max(len(next(iter(i[field].values()))) for i in instances)
        batch[field] = {}
        for k in keys:
            pad = -1 if 'copy_masks' in k else padding_value
            rows = np.full((len(instances), n), pad, dtype=np.int64)
            for j, inst in enumerate(instances):
                v = inst[field][k]
                rows[j, :len(v)] = v
            batch[field][k] = host(rows)
    batch['image_u8'] = host(np.stack([i['image'] for i in instances]))
    for field in ('face_embeds', 'obj_embeds'):
        if field not in instances[0]:
            continue
        arrs = [np.asarray(i[field], dtype=np.float32) for i in instances]
        rows = max(a.shape[0] for a in arrs)
        dim = max(a.shape[1] for a in arrs)
        out = np.full((len(arrs), rows, dim), np.nan, dtype=np.float32)
        for j, a in enumerate(arrs):
            out[j, :a.shape[0], :a.shape[1]] = a
        batch[field] = host(out)
    batch['metadata'] = [i['metadata'] for i in instances]
    return batch


def to_device(host_batch, device='cpu'):
    """The host batch of `collate_host` on `device`, pixels through ToTensor + Normalize (one HIP kernel on the GPU):
    the kwargs of Model.forward (transformer_faces_objects.py:67-76)."""
    nb = torch.device(device).type == 'cuda'
    batch = {}
    for k, v in host_batch.items():
        if k == 'image_u8':
            batch['image'] = normalize_images(v.to(device, non_blocking=nb) if nb else v.numpy(), device)
        elif isinstance(v, dict):
            batch[k] = {kk: vv.to(device, non_blocking=nb) for kk, vv in v.items()}
        elif torch.is_tensor(v):
            batch[k] = v.to(device, non_blocking=nb)
        else:
            batch[k] = v
    return batch


def collate(instances, device='cpu', padding_value=1):
    """list of reader instances -> the kwargs of Model.forward (transformer_faces_objects.py:67-76)."""
    return to_device(collate_host(instances, padding_value), device)


@DataIterator.register('bucket')
class BucketIterator(DataIterator):
    def __init__(self, sorting_keys, padding_noise=0.1, biggest_batch_first=False, batch_size=32, instances_per_epoch=None,
                 max_instances_in_memory=None, cache_instances=False, track_epoch=False, maximum_samples_per_batch=None,
                 skip_smaller_batches=False, seed=1234):
        self.sorting_keys = [tuple(k) for k in sorting_keys]
        self.padding_noise, self.biggest_batch_first = padding_noise, biggest_batch_first
        self.batch_size, self.instances_per_epoch = batch_size, instances_per_epoch
        self.max_instances_in_memory = max_instances_in_memory
        self.maximum_samples_per_batch = tuple(maximum_samples_per_batch) if maximum_samples_per_batch else None
        self.skip_smaller_batches = skip_smaller_batches
        self.rng = random.Random(seed)
        self._cursor = None

    # ---- chunks of the (possibly lazy) instance stream
    def _memory_sized_lists(self, instances):
        it = iter(instances) if self.instances_per_epoch is None else self._take_epoch(instances)
        size = self.max_instances_in_memory or self.instances_per_epoch
        if size is None:
            yield list(it)
            return
        chunk = []
        for inst in it:
            chunk.append(inst)
            if len(chunk) == size:
                yield chunk
                chunk = []
        if chunk:
            yield chunk

    def _take_epoch(self, instances):
        """instances_per_epoch: an epoch is the next n instances of an endless pass over the data (the cursor survives
        between epochs)."""
        if self._cursor is None:
            self._cursor = iter(instances)
        for _ in range(self.instances_per_epoch):
            try:
                yield next(self._cursor)
            except StopIteration:
                self._cursor = iter(instances)
                try:
                    yield next(self._cursor)
                except StopIteration:
                    return

    def _sorted(self, instances):
        def key(inst):
            out = []
            for field, name in self.sorting_keys:
                n = padding_length(inst, field, name)
                out.append(n * (1.0 + self.rng.uniform(-self.padding_noise, self.padding_noise)))
            return out
        return [instances[j] for _, j in sorted((key(i), j) for j, i in enumerate(instances))]

    def _fits(self, batch):
        if self.maximum_samples_per_batch is None:
            return True
        key, limit = self.maximum_samples_per_batch
        longest = max(max(padding_length(i, f, key) for f in ('context', 'caption') if f in i) for i in batch)
        return longest * len(batch) <= limit

    def _batches(self, instances, shuffle):
        for chunk in self._memory_sized_lists(instances):
            batches, cur = [], []
            for inst in self._sorted(chunk):
                if len(cur) == self.batch_size or (cur and not self._fits(cur + [inst])):
                    batches.append(cur)
                    cur = []
                cur.append(inst)
            if cur and not (self.skip_smaller_batches and len(cur) < self.batch_size):
                batches.append(cur)
            move_to_front = self.biggest_batch_first and len(batches) > 1
            if move_to_front:
                last, penultimate = batches.pop(), batches.pop()
            if shuffle:
                self.rng.shuffle(batches)
            if move_to_front:
                batches.insert(0, penultimate)
                batches.insert(0, last)
            yield from batches

    def __call__(self, instances, num_epochs=1, shuffle=True, device='cpu'):
        for _ in range(num_epochs):
            for b in self._batches(instances, shuffle):
                yield collate(b, device)

    def get_num_batches(self, instances):
        n = self.instances_per_epoch or len(list(instances))
        return (n + self.batch_size - 1) // self.batch_size
