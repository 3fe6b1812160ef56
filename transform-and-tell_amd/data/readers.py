"""Dataset readers of the hot path under the reference's registration names
(tell/data/dataset_readers/nytimes_faces_ner_matched.py:35-36 `nytimes_faces_ner_matched`, nytimes.py `nytimes`,
goodnews_flattened_glove.py `goodnews_flattened_glove`) and constructor keys (config.yaml:1-21).  The source of the
articles is an on-disk shard directory (data/shards.py) instead of MongoDB + a JPEG directory; with no shard directory
the readers fall back to synthetic samples of the same contract, so every expt/ config still instantiates and runs.

An instance is a dict with the reader's field names (nytimes_faces_ner_matched.py:208-225):
    context / caption  {'roberta': [ids], 'roberta_copy_masks': [0/1]}     image  uint8 [224,224,3]
    face_embeds  float32 [F,512] or the empty field [1,0]                   obj_embeds  float32 [O,2048] or [1,0]
    metadata  dict
Batches are assembled by data/iterators.py (padding: ids 1, copy masks -1, arrays NaN - the ArrayFields' padding_value)."""
import numpy as np

from .indexers import TokenIndexer
from .shards import read_shard, shard_paths
from .synthetic import DatasetReader, synthetic_batch


def _indexers_from_params(token_indexers):
    out = {}
    for name, p in (token_indexers or {}).items():
        if isinstance(p, dict):
            p = dict(p)
            out[name] = TokenIndexer.by_name(p.pop('type'))(**p)
        else:
            out[name] = p
    return out


@DatasetReader.register('nytimes_faces_ner_matched')
@DatasetReader.register('nytimes')
class NYTimesFacesNERMatchedReader(DatasetReader):
    EMPTY = np.zeros((1, 0), dtype=np.float32)          # `np.array([[]])` of :166,182,186

    def __init__(self, tokenizer=None, token_indexers=None, image_dir=None, mongo_host='localhost', mongo_port=27017,
                 use_caption_names=True, use_objects=False, n_faces=None, lazy=True, shard_dir=None,
                 synthetic_samples=64, seed=1234, **unused):
        self._token_indexers = _indexers_from_params(token_indexers)
        self.index_name = next(iter(self._token_indexers), 'roberta')
        self.image_dir, self.shard_dir = image_dir, shard_dir
        self.use_caption_names, self.use_objects, self.n_faces = use_caption_names, use_objects, n_faces
        self.synthetic_samples = synthetic_samples
        self.rs = np.random.RandomState(seed)            # nytimes_faces_ner_matched.py:73-74

    # ---- :81-190 (the Mongo query / paragraph selection happened when the shard was written)
    def _read(self, split):
        if split not in ('train', 'valid', 'test'):
            raise ValueError('Unknown split: %s' % split)
        if self.shard_dir is None:
            yield from self._synthetic(split)
            return
        paths = shard_paths(self.shard_dir, split)
        if not paths:
            raise FileNotFoundError('no %s-*.npz shards in %r' % (split, self.shard_dir))
        for p in [paths[i] for i in self.rs.permutation(len(paths))]:
            for s in read_shard(p):
                faces = s['face_embeds']
                if self.n_faces is not None:                                   # :125-130
                    n_persons = self.n_faces
                elif self.use_caption_names:
                    # PERSON names of the caption: recorded by the shard writer, or already applied by it (shards.py)
                    n_persons = s.get('n_person_names')
                    if n_persons is None:
                        if len(faces) > 4:
                            raise ValueError('shard %s: %d faces for one sample and no n_person_names - with '
                                             'use_caption_names the writer must record the name count or pre-trim'
                                             % (p, len(faces)))
                        n_persons = len(faces)
                else:
                    n_persons = 4
                faces = faces[:n_persons] if n_persons else faces[:0]
                obj = s.get('obj_embeds') if self.use_objects else None
                yield self.ids_to_instance(s['context_ids'], s['caption_ids'], s['image'], faces, obj, s['metadata'],
                                           s.get('context_copy'), s.get('caption_copy'))

    read = _read

    def _synthetic(self, split):
        b = synthetic_batch(self.synthetic_samples, 512, 33, True, seed=1234 + {'train': 0, 'valid': 1, 'test': 2}[split],
                            variable=True)
        img = (b['image'].clamp(-2, 2) * 50 + 128).to('cpu').numpy().astype(np.uint8).transpose(0, 2, 3, 1)
        for i in range(self.synthetic_samples):
            ctx = b['context']['roberta'][i]
            cap = b['caption']['roberta'][i]
            faces = b['face_embeds'][i].numpy()
            objs = b['obj_embeds'][i].numpy()
            yield self.ids_to_instance(ctx[ctx != 1].numpy(), cap[cap != 1].numpy(), img[i],
                                       faces[~np.isnan(faces).any(1)], objs[~np.isnan(objs).any(1)] if self.use_objects
                                       else None, {'caption': '', 'context': '', 'web_url': '', 'image_path': '',
                                                   'image_pos': 0})

    # ---- :192-227 from text (needs the BPE files) ...
    def article_to_instance(self, paragraphs, named_entities, image, caption, image_path, web_url, pos, face_embeds,
                            obj_feats):
        context = '\n'.join(paragraphs).strip()
        idx = self._token_indexers[self.index_name]
        ctx_ids, ctx_copy = idx.encode(context)
        cap_ids, cap_copy = idx.encode(caption)
        meta = {'context': context, 'caption': caption, 'names': named_entities, 'web_url': web_url,
                'image_path': image_path, 'image_pos': pos}
        return self.ids_to_instance(ctx_ids, cap_ids, np.asarray(image, dtype=np.uint8), np.asarray(face_embeds),
                                    None if obj_feats is None else np.asarray(obj_feats), meta, ctx_copy, cap_copy)

    # ---- ... or from ids (shards are pre-tokenised)
    def ids_to_instance(self, context_ids, caption_ids, image, face_embeds, obj_embeds, metadata, context_copy=None,
                        caption_copy=None):
        n = self.index_name

        def field(ids, copy):
            ids = np.asarray(ids).tolist()
            return {n: ids, n + '_copy_masks': np.asarray(copy).tolist() if copy is not None else [0] * len(ids)}
        faces = np.asarray(face_embeds, dtype=np.float32)
        inst = {'context': field(context_ids, context_copy), 'caption': field(caption_ids, caption_copy),
                'image': np.asarray(image, dtype=np.uint8),
                'face_embeds': faces.reshape(-1, 512) if faces.size else self.EMPTY, 'metadata': metadata}
        if obj_embeds is not None:
            objs = np.asarray(obj_embeds, dtype=np.float32)
            inst['obj_embeds'] = objs.reshape(-1, 2048) if objs.size else self.EMPTY
        return inst


@DatasetReader.register('goodnews_flattened_glove')
class FlattenedGloveGoodNewsReader(NYTimesFacesNERMatchedReader):
    """goodnews_flattened_glove.py:24-25 (expt/goodnews/1_lstm_glove, 2_transformer_glove): image + caption ids; the article
    reaches the model as GloVe vectors (`context_vectors`, spaCy - absent here), so the instance carries the article ids
    only for bucketing."""

    def __init__(self, tokenizer=None, token_indexers=None, image_dir=None, mongo_host='localhost', mongo_port=27017,
                 eval_limit=5120, lazy=True, shard_dir=None, **kw):
        super().__init__(tokenizer, token_indexers, image_dir, mongo_host, mongo_port, lazy=lazy, shard_dir=shard_dir, **kw)
        self.eval_limit = eval_limit
