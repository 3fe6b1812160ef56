"""Shard builder: the reference's NYTimes reader run ONCE, offline, over its own sources - a MongoDB `nytimes` database
(or any iterable of its article documents) and the directory of 224x224 JPEGs - writing the pre-indexed shards that
`data/readers.py` serves to the trainer (data/shards.py: token ids, uint8 pixels, ragged face / object arrays).

Field logic restated from tell/data/dataset_readers/nytimes_faces_ner_matched.py:81-190 (NYTimesFacesNERMatchedReader):
  * one sample per image position of an article (:107); a sample is dropped when its caption is empty (:120-121) or its
    image file cannot be opened (:160-163);
  * context = headline (:109-117), the FIRST paragraph of the article (:136-140), then the paragraphs before and after
    the image, nearest first, until 510 BPE tokens of headline + surrounding paragraphs are collected or the article is
    exhausted (:142-158) - in the order headline, first paragraph, before (document order), after (:172);
  * the number of PERSON names in the caption section (:127-128, :240-250) is RECORDED (`n_person_names`): the reader
    applies `n_faces` / `use_caption_names` / the default of 4 when it serves the sample (readers.py), so one set of shards
    serves every expt/ configuration; face embeddings are stored in the database's order (largest face first, :168-170);
  * object features from the `objects` collection by image hash (:175-186); a missing document is the empty array.
Tokenisation: the RoBERTa byte-level BPE + fairseq dictionary of data/bpe.py through RobertaTokenIndexer (<s> ... </s>,
truncation to max_len 512) - the files (encoder.json, vocab.bpe, dict.txt) are not part of this repository.
This module is host-side tooling (pymongo and Pillow are only imported where used); nothing on the GPU path imports it."""
import os

import numpy as np

from .shards import write_shard


def _named(section, labels):
    """:228-250 - the texts of the section's named entities with one of `labels`."""
    return {ner['text'] for ner in section.get('named_entities', ()) if ner['label'] in labels}


def _outward(pos, first, n):
    """Section indices around the image at `pos`, one ring per step: (pos - r, pos + r) for r = 1, 2, ... until both sides
    have left the article body (indices <= first on the left, >= n on the right).  The reference walks two cursors
    (:130-158); this is the same visiting order as a generator."""
    r = 1
    while True:
        yield pos - r, pos + r
        if pos - r - 1 <= first and pos + r + 1 >= n:
            return
        r += 1


def article_samples(article, n_tokens, budget=510):
    """The (position, caption, paragraphs, named entities, caption section) tuples of one article document (:104-172):
    context = headline + the article's first paragraph + the paragraphs nearest to the image, taken ring by ring
    (one to the left, one to the right) until `budget` BPE tokens are reached - the test is made once per RING, and the
    first paragraph is not counted (both as the reference has it).
    n_tokens(text) -> number of BPE tokens (the reference counts with fairseq's roberta.bpe, :252-260)."""
    sections = article['parsed_section']
    labels = ('PERSON', 'ORG', 'GPE')
    is_par = [sec['type'] == 'paragraph' for sec in sections]
    # index of the first paragraph; an article without one behaves like the reference's fall-through (its loop
    # variable ends on the last section)
    first = is_par.index(True) if True in is_par else len(sections) - 1
    title = article.get('headline', {}).get('main', '').strip() if 'main' in article.get('headline', {}) else ''
    for pos in article['image_positions']:
        caption = sections[pos]['text'].strip()
        if not caption:
            continue
        head = [title] if title else []
        # (:115-116 calls set.union without keeping the result: the headline's entities are NOT collected)
        spent = n_tokens(title) if title else 0
        picked = [first] if True in is_par else []
        for ring in _outward(pos, first, len(sections)):
            for idx in ring:
                if first < idx < len(sections) and is_par[idx]:
                    picked.append(idx)
                    spent += n_tokens(sections[idx]['text'])
            if spent >= budget:
                break
        picked.sort()                                            # [first] + left side + right side, each in article order
        named = set()
        for idx in picked:
            named |= _named(sections[idx], labels)
        yield pos, caption, head + [sections[idx]['text'] for idx in picked], sorted(named), sections[pos]


def load_image(path):
    """-> uint8 [224, 224, 3] or None when the file is missing / unreadable (:160-163).  The reference's images are
    already 224x224 on disk (its reader only applies ToTensor + Normalize, :66-68); anything else is an error here."""
    from PIL import Image
    try:
        with Image.open(path) as im:
            a = np.asarray(im.convert('RGB'), dtype=np.uint8)
    except (FileNotFoundError, OSError):
        return None
    if a.shape != (224, 224, 3):
        raise ValueError('%s: %r - the reader expects the 224x224 crops of the reference\'s image directory' % (path, a.shape))
    return a


def build_shards(articles, image_dir, out_dir, split, indexer, objects=None, use_objects=True, shard_size=128,
                 max_objects=64, shuffle_seed=1234):
    """articles: iterable of article documents of ONE split (dicts with _id, parsed_section, image_positions, headline,
    web_url - the projection of :98-102); objects: mapping image hash -> object document (or a callable), as
    db.objects.find_one would return it.  Writes out_dir/<split>-NNNNN.npz; -> (samples written, samples skipped).
    shuffle_seed: the reference shuffles the article ids of a split with RandomState(1234) (:95); None keeps the order."""
    articles = sorted(articles, key=lambda a: a['_id'])
    if shuffle_seed is not None:
        order = np.arange(len(articles))
        np.random.RandomState(shuffle_seed).shuffle(order)
        articles = [articles[i] for i in order]
    bpe = indexer.bpe.bpe
    n_tokens = lambda text: len(bpe.encode(text))            # noqa: E731
    lookup = objects if callable(objects) else (lambda h: (objects or {}).get(h))
    buf, written, skipped, shard = [], 0, 0, 0

    def flush():
        nonlocal buf, shard
        if buf:
            write_shard(os.path.join(out_dir, '%s-%05d.npz' % (split, shard)), buf)
            shard += 1
            buf = []
    for article in articles:
        for pos, caption, paragraphs, named, section in article_samples(article, n_tokens):
            image_path = os.path.join(image_dir, '%s.jpg' % section['hash'])
            image = load_image(image_path)
            if image is None:
                skipped += 1
                continue
            faces = np.zeros((0, 512), np.float32)
            if 'facenet_details' in section:
                faces = np.asarray(section['facenet_details']['embeddings'], dtype=np.float32).reshape(-1, 512)
            context = '\n'.join(paragraphs).strip()            # :192
            sample = {'context_ids': indexer.encode(context)[0], 'caption_ids': indexer.encode(caption)[0],
                      'image': image, 'face_embeds': faces,
                      'n_person_names': len(_named(section, ('PERSON',))),
                      'metadata': {'context': context, 'caption': caption, 'names': named,
                                   'web_url': article.get('web_url', ''), 'image_path': image_path, 'image_pos': pos}}
            if use_objects:
                doc = lookup(section['hash'])
                feats = np.asarray(doc['object_features'] if doc is not None else [], dtype=np.float32).reshape(-1, 2048)
                sample['obj_embeds'] = feats[:max_objects]
            buf.append(sample)
            written += 1
            if len(buf) == shard_size:
                flush()
    flush()
    return written, skipped


def build_from_mongo(image_dir, out_dir, indexer, splits=('train', 'valid', 'test'), host='localhost', port=27017,
                     use_objects=True, **kw):
    """The reference's own source (:58-60, :88-102): the `nytimes` database of a running MongoDB."""
    from pymongo import MongoClient
    db = MongoClient(host=host, port=port).nytimes
    projection = ['_id', 'parsed_section.type', 'parsed_section.text', 'parsed_section.hash',
                  'parsed_section.facenet_details', 'parsed_section.named_entities', 'image_positions', 'headline', 'web_url']
    out = {}
    for split in splits:
        docs = db.articles.find({'split': split}, projection=projection)
        objects = (lambda h: db.objects.find_one({'_id': h})) if use_objects else None
        out[split] = build_shards(docs, image_dir, out_dir, split, indexer, objects=objects, use_objects=use_objects, **kw)
    return out
