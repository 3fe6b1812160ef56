"""CPU: the data plane - RoBERTa byte-level BPE + indexer, shard format, readers under the
reference's registration names, bucket iterator and the batch tensor contract of Model.forward."""
import json
import os

import numpy as np
import pytest
import torch


def _write_vocab(d):
    """A small synthetic RoBERTa vocabulary: all 256 byte symbols + a handful of merges, fairseq dict.txt on top."""
    from tell_amd.data.bpe import bytes_to_unicode
    chars = list(bytes_to_unicode().values())
    merges = [('Ġ', 't'), ('Ġt', 'h'), ('Ġth', 'e'), ('h', 'e'), ('l', 'l'), ('he', 'll'), ('hell', 'o'),
              ('Ġ', 'M'), ('ĠM', 'i'), ('l', 'a'), ('ĠMi', 'la'), ('ĠMila', 'n')]
    tokens = chars + [a + b for a, b in merges]
    enc = {t: i for i, t in enumerate(tokens)}
    with open(os.path.join(d, 'encoder.json'), 'w') as f:
        json.dump(enc, f)
    with open(os.path.join(d, 'vocab.bpe'), 'w', encoding='utf-8') as f:
        f.write('#version: 0.2\n' + '\n'.join('%s %s' % m for m in merges) + '\n')
    order = list(reversed(range(len(tokens))))                 # fairseq orders by frequency: any permutation
    with open(os.path.join(d, 'dict.txt'), 'w') as f:
        f.write('\n'.join('%d %d' % (i, 1000 - k) for k, i in enumerate(order)) + '\n')
    return enc, order


def test_byte_bpe_known_answers_and_round_trip(tmp_path):
    from tell_amd.data.bpe import RobertaBPE
    enc, order = _write_vocab(str(tmp_path))
    rb = RobertaBPE(str(tmp_path))
    # 'hello' merges h+e, l+l, he+ll, hell+o -> one token; ' the' -> 'Gthe'; ' them' -> 'Gthe' + 'm'
    assert rb.bpe.encode('hello') == [enc['hello']]
    assert rb.bpe.encode('hello the them') == [enc['hello'], enc['Ġthe'], enc['Ġthe'], enc['m']]
    assert rb.bpe.pretokenize("Tomas Maier, autumn/winter 2014,\n in Milan.") == \
        ['Tomas', ' Maier', ',', ' autumn', '/', 'winter', ' 2014', ',', '\n', ' in', ' Milan', '.']
    for text in ['hello the  world\n\nnew para', 'café — naïve 中文 \U0001f600', " it's 12,345.67 "]:
        assert rb.bpe.decode(rb.bpe.encode(text)) == text
    # fairseq ids: <s>=0 ... </s>=2, symbol i of dict.txt -> 4 + its line
    ids = rb.encode_ids('hello the')
    assert ids[0] == 0 and ids[-1] == 2
    assert ids[1] == 4 + order.index(enc['hello']) and ids[2] == 4 + order.index(enc['Ġthe'])
    assert rb.decode(ids[1:]) == 'hello the'                   # stops at </s>
    assert len(rb.encode_ids('x ' * 600, max_len=512)) == 512


GPT2_CASES = ["I'm I'M  it's 12,345.67  end  ", "a\n\nb\t c", "naïve café 中文 \U0001f600!!! ...x", "don't they'll we've I'd he's you're",
              "  leading and trailing\n", "tabs\t\tand\r\nCRLF", "x" * 40 + " 3.14159e-10 #hash_tag @user http://a.b/c?d=e",
              "\u00a0non-breaking\u2003em space", "\x00\x7f\xad bytes"]


def test_gpt2_pretokenizer_byte_map_and_merges_against_the_tokenizers_library(tmp_path):
    """Known answers for the pre-tokenizer regex and the byte fallback (reference call site
    roberta_indexer.py:89-109 -> fairseq's GPT-2 encoder).  The independent implementation is Hugging Face's
    `tokenizers` (Rust): its ByteLevel pre-tokenizer is the published GPT-2 regex + byte -> unicode map, its BPE model the
    ranked-merge loop.  Contractions are case sensitive, a run of spaces leaves its last space to the next word, letters
    / digits / other split, any byte without a merge stays a single byte symbol (space -> U+0120, newline -> U+010A, the
    three non-printable ranges map to U+0100..U+0143)."""
    tokenizers = pytest.importorskip('tokenizers')
    from tokenizers import Tokenizer, models, pre_tokenizers
    from tell_amd.data.bpe import RobertaBPE, bytes_to_unicode
    b2u = bytes_to_unicode()
    assert len(b2u) == 256 and len(set(b2u.values())) == 256
    assert (b2u[0x20], b2u[0x0a], b2u[0x09], b2u[0x00], b2u[0x7f], b2u[0xa0], b2u[0xad]) == \
        ('\u0120', '\u010a', '\u0109', '\u0100', '\u0121', '\u0142', '\u0143')
    assert all(b2u[b] == chr(b) for b in list(range(33, 127)) + list(range(161, 173)) + list(range(174, 256)))
    enc, _ = _write_vocab(str(tmp_path))
    rb = RobertaBPE(str(tmp_path))
    ref_pre = pre_tokenizers.ByteLevel(add_prefix_space=False, use_regex=True)
    with open(os.path.join(str(tmp_path), 'vocab.bpe'), encoding='utf-8') as f:
        merges = [tuple(ln.split()) for ln in f.read().split('\n')[1:] if ln.strip()]
    ref = Tokenizer(models.BPE(vocab=enc, merges=merges))
    ref.pre_tokenizer = ref_pre
    for text in GPT2_CASES:
        mine = [''.join(b2u[b] for b in tok.encode('utf-8')) for tok in rb.bpe.pretokenize(text)]
        assert mine == [t for t, _ in ref_pre.pre_tokenize_str(text)], text
        assert rb.bpe.encode(text) == ref.encode(text).ids, text
        assert rb.bpe.decode(rb.bpe.encode(text)) == text
    assert rb.bpe.pretokenize("I'm I'M") == ['I', "'m", ' I', "'", 'M']
    assert rb.bpe.pretokenize('a  b   ') == ['a', ' ', ' b', '   ']
    assert rb.bpe.pretokenize('abc123!?') == ['abc', '123', '!?']


def test_roberta_indexer_contract(tmp_path):
    """roberta_indexer.py:89-109,185-200: <s> ... </s>, truncation to max_len, entity copy masks, padding 1 / -1."""
    from tell_amd.data import RobertaTokenIndexer, TokenIndexer
    from tell_amd.data.bpe import RobertaBPE
    _write_vocab(str(tmp_path))
    assert TokenIndexer.by_name('roberta') is RobertaTokenIndexer
    idx = RobertaTokenIndexer(model_name='roberta-base', namespace='bpe', padding_on_right=True, padding_value=1,
                              max_len=16, bpe=RobertaBPE(str(tmp_path)))

    class Ent:
        def __init__(self, s, e):
            self.start_char, self.end_char = s, e

    class Doc:
        ents = [Ent(10, 15)]                                   # "Milan" in "hello the Milan x"
    text = 'hello the Milan x'
    ids, copy = idx.encode(text, Doc())
    assert ids[0] == 0 and ids[-1] == 2 and len(ids) == len(copy)
    assert copy == [0, 0, 0, 1, 0, 0, 0]                       # <s> hello Gthe GMilan G x </s>: only ' Milan' is inside
    out = idx.tokens_to_indices(text.split(' '), None, 'roberta')
    assert out['roberta'] == idx.encode(text)[0] and set(out) == {'roberta', 'roberta_copy_masks'}
    long_ids, long_copy = idx.encode('x ' * 100)
    assert len(long_ids) == 16 and long_ids[-1] == 2 and len(long_copy) == 16
    padded = idx.as_padded_tensor({'roberta': ids, 'roberta_copy_masks': copy}, {'roberta': 10, 'roberta_copy_masks': 10})
    assert padded['roberta'].dtype == torch.long and padded['roberta'].tolist()[-3:] == [1, 1, 1]
    assert padded['roberta_copy_masks'].tolist()[-3:] == [-1, -1, -1]
    with pytest.raises(FileNotFoundError):                    # no silent fallback when the BPE files are missing
        RobertaBPE(str(tmp_path / 'nowhere'))


def _samples(n, seed=0, with_obj=True):
    g = np.random.RandomState(seed)
    out = []
    for i in range(n):
        L, T, F, O = g.randint(20, 200), g.randint(5, 30), g.randint(0, 5), g.randint(0, 9)
        s = {'context_ids': np.r_[0, g.randint(4, 50265, L - 2), 2], 'caption_ids': np.r_[0, g.randint(4, 50265, T - 2), 2],
             'image': g.randint(0, 256, (224, 224, 3)).astype(np.uint8), 'face_embeds': g.randn(F, 512).astype(np.float32),
             'metadata': {'caption': 'c%d' % i, 'context': 'x', 'web_url': 'u', 'image_path': 'p%d.jpg' % i, 'image_pos': 0}}
        if with_obj:
            s['obj_embeds'] = np.abs(g.randn(O, 2048)).astype(np.float32)
        out.append(s)
    return out


def test_shards_reader_and_batch_contract(tmp_path):
    """write -> read round trip, reader under the reference's name and constructor keys, the tensors Model.forward gets
    (ids right-padded with 1, <s>=0 first, </s>=2 last real token, copy masks -1, faces <= 4 rows and
    objects NaN-padded, the empty [1,0] field, normalised image)."""
    from tell_amd.data import BucketIterator, DatasetReader, collate, read_shard, write_shard
    samples = _samples(11)
    samples[3]['face_embeds'] = np.zeros((0, 512), np.float32)
    write_shard(str(tmp_path / 'train-00000.npz'), samples[:6])
    write_shard(str(tmp_path / 'train-00001.npz'), samples[6:])
    back = read_shard(str(tmp_path / 'train-00000.npz'))
    assert len(back) == 6
    for a, b in zip(samples[:6], back):
        assert np.array_equal(a['context_ids'], b['context_ids']) and np.array_equal(a['image'], b['image'])
        assert np.array_equal(a['face_embeds'], b['face_embeds']) and a['metadata'] == b['metadata']
    reader = DatasetReader.by_name('nytimes_faces_ner_matched')(
        tokenizer={'type': 'word'}, token_indexers={'roberta': {'type': 'roberta', 'model_name': 'roberta-base',
                                                                'namespace': 'bpe', 'padding_on_right': True,
                                                                'padding_value': 1, 'max_len': 512}},
        image_dir='unused', lazy=True, use_caption_names=False, use_objects=True, shard_dir=str(tmp_path))
    inst = list(reader._read('train'))
    assert len(inst) == 11 and set(inst[0]) == {'context', 'caption', 'image', 'face_embeds', 'obj_embeds', 'metadata'}
    empties = [i for i in inst if i['face_embeds'].shape == (1, 0)]
    assert len(empties) >= 1                                               # `np.array([[]])`, reader :166
    with pytest.raises(ValueError):
        list(reader._read('dev'))
    batch = collate(inst[:5])
    ctx, cap = batch['context']['roberta'], batch['caption']['roberta']
    assert ctx.dtype == torch.long and (ctx[:, 0] == 0).all()
    for row, i in zip(ctx, inst[:5]):
        n = len(i['context']['roberta'])
        assert row[n - 1] == 2 and (row[n:] == 1).all()
    assert (batch['caption']['roberta_copy_masks'][cap == 1] == -1).all()
    assert batch['image'].shape == (5, 3, 224, 224) and batch['image'].dtype == torch.float32
    want = (torch.from_numpy(np.array(inst[0]['image'])).permute(2, 0, 1).float() / 255 -
            torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1)) / torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1)
    torch.testing.assert_close(batch['image'][0], want)
    fe, oe = batch['face_embeds'], batch['obj_embeds']
    assert fe.shape[0] == 5 and fe.shape[1] <= 4 and fe.shape[2] == 512 and oe.shape[2] == 2048
    for j, i in enumerate(inst[:5]):
        f = i['face_embeds']
        nf = 0 if f.shape == (1, 0) else f.shape[0]
        assert torch.isnan(fe[j, nf:]).all() and not torch.isnan(fe[j, :nf]).any()
    all_empty = collate([e for e in empties for _ in range(2)][:2])
    assert all_empty['face_embeds'].shape == (2, 1, 0)                      # the kdim == 0 branch of the attention
    # the whole pipeline, iterator of config.yaml:99-110
    it = BucketIterator(sorting_keys=[['context', 'num_tokens'], ['caption', 'num_tokens']], batch_size=4,
                        max_instances_in_memory=8192, biggest_batch_first=False, instances_per_epoch=8,
                        maximum_samples_per_batch=['num_tokens', 16384])
    b1 = list(it(inst, num_epochs=1, shuffle=True))
    assert sum(b['context']['roberta'].shape[0] for b in b1) == 8          # one epoch = 8 instances
    b2 = list(it(inst, num_epochs=1, shuffle=True))
    assert sum(b['context']['roberta'].shape[0] for b in b2) == 8          # the cursor continues (3 left + 5 new)


def test_many_live_shards_hold_one_descriptor_each(tmp_path):
    """A validation / test pass keeps every instance of the split alive (BucketIterator materialises list(it)): views of
    more than 400 memory-mapped shards at once under RLIMIT_NOFILE = 256 - one mapping, and no lingering descriptor, per
    shard (three np.memmap objects per shard died here with EMFILE 'Too many open files')."""
    import resource
    import numpy as np
    from tell_amd.data import shards
    rng = np.random.RandomState(0)
    base = dict(context_ids=[0, 5, 6, 2], caption_ids=[0, 7, 2], image=np.zeros((224, 224, 3), np.uint8),
                face_embeds=rng.rand(1, 512).astype(np.float32), obj_embeds=rng.rand(2, 2048).astype(np.float32), metadata={})
    first = str(tmp_path / 'test-00000.npz')
    shards.write_shard(first, [base])
    import shutil
    for i in range(1, 420):
        shutil.copyfile(first, str(tmp_path / ('test-%05d.npz' % i)))
    soft, hard = resource.getrlimit(resource.RLIMIT_NOFILE)
    resource.setrlimit(resource.RLIMIT_NOFILE, (256, hard))
    try:
        live = []
        for path in shards.shard_paths(str(tmp_path), 'test'):
            live.extend(shards.read_shard(path))
        assert len(live) == 420
        assert all(isinstance(s['image'].base, (np.ndarray, np.memmap)) or s['image'].base is not None for s in live)
        assert float(live[-1]['obj_embeds'].sum()) == float(np.asarray(base['obj_embeds']).sum())
        assert live[7]['context_ids'].tolist() == [0, 5, 6, 2]
    finally:
        resource.setrlimit(resource.RLIMIT_NOFILE, (soft, hard))


def test_reader_limits_faces_to_caption_person_names(tmp_path):
    """nytimes_faces_ner_matched.py:125-130,168-170: n_faces wins; else with use_caption_names the number of PERSON names
    in the caption (recorded by the shard writer as n_person_names) limits the faces - 0 names = the empty field; a
    shard without the count is taken as pre-trimmed, and more than 4 faces in it is an error, not a silent truncation."""
    from tell_amd.data import DatasetReader, write_shard
    samples = _samples(6, seed=3, with_obj=False)
    names = [0, 1, 2, 4, 3, 1]
    for s, n in zip(samples, names):
        s['face_embeds'] = np.random.RandomState(n).randn(4, 512).astype(np.float32)
        s['n_person_names'] = n
    write_shard(str(tmp_path / 'a' / 'train-00000.npz'), samples)
    mk = lambda d, **kw: DatasetReader.by_name('nytimes_faces_ner_matched')(shard_dir=str(tmp_path / d), seed=0, **kw)  # noqa: E731
    by_img = {s['image'].tobytes(): n for s, n in zip(samples, names)}
    for inst in mk('a', use_caption_names=True)._read('train'):
        n = by_img[inst['image'].tobytes()]
        assert inst['face_embeds'].shape == ((n, 512) if n else (1, 0))
    for inst in mk('a', use_caption_names=True, n_faces=2)._read('train'):       # n_faces overrides (:125-126)
        assert inst['face_embeds'].shape == (2, 512)
    for inst in mk('a', use_caption_names=False)._read('train'):                 # neither: the top 4 faces (:129-130)
        assert inst['face_embeds'].shape == (4, 512)
    for s in samples:
        del s['n_person_names']
        s['face_embeds'] = np.zeros((5, 512), np.float32)
    write_shard(str(tmp_path / 'b' / 'train-00000.npz'), samples)
    with pytest.raises(ValueError):
        list(mk('b', use_caption_names=True)._read('train'))


def test_bucket_iterator_semantics():
    from tell_amd.data import BucketIterator, NYTimesFacesNERMatchedReader
    reader = NYTimesFacesNERMatchedReader(use_objects=True, synthetic_samples=40)
    inst = list(reader._read('valid'))
    it = BucketIterator(sorting_keys=[['context', 'num_tokens']], batch_size=16, padding_noise=0.0,
                        maximum_samples_per_batch=['num_tokens', 2048])
    batches = list(it._batches(inst, shuffle=False))
    assert sum(len(b) for b in batches) == 40
    lens = [len(i['context']['roberta']) for b in batches for i in b]
    assert lens == sorted(lens)                                            # bucketed by article length
    for b in batches:
        longest = max(max(len(i['context']['roberta']), len(i['caption']['roberta'])) for i in b)
        assert len(b) <= 16 and longest * len(b) <= 2048
    big = BucketIterator(sorting_keys=[['context', 'num_tokens']], batch_size=16, padding_noise=0.0, biggest_batch_first=True)
    bb = list(big._batches(inst, shuffle=False))
    assert max(len(i['context']['roberta']) for i in bb[0]) == max(lens)   # the longest batch is run first


REF_CFG = '/root/reference/expt/nytimes/9_transformer_objects/config.yaml'


@pytest.mark.skipif(not os.path.exists(REF_CFG), reason='reference tree only exists in the authoring container')
def test_reference_config_data_sections_instantiate_unmodified():
    from tell_amd import config
    from tell_amd.data import BucketIterator, NYTimesFacesNERMatchedReader, RobertaVocabulary
    for name in ('9_transformer_objects', '5_transformer_roberta', '8_transformer_faces'):
        p = config.yaml_to_params(REF_CFG.replace('9_transformer_objects', name))
        reader = config.reader_from_params(p['dataset_reader'])
        assert isinstance(reader, NYTimesFacesNERMatchedReader) or type(reader).__name__.startswith('NYTimes')
        assert isinstance(config.iterator_from_params(p['iterator']), BucketIterator)
        assert isinstance(config.iterator_from_params(p['validation_iterator']), BucketIterator)
        assert isinstance(config.vocabulary_from_params(p['vocabulary']), RobertaVocabulary)
    r9 = config.reader_from_params(config.yaml_to_params(REF_CFG)['dataset_reader'])
    assert r9.use_objects is True and next(iter(r9._token_indexers.values()))._max_len == 512
    gp = config.yaml_to_params(REF_CFG.replace('nytimes', 'goodnews').replace('9_transformer_objects', '1_lstm_glove'))
    assert type(config.reader_from_params(gp['dataset_reader'])).__name__ == 'FlattenedGloveGoodNewsReader'


def test_bleu_scorer_known_answers():
    """Hand-computed BLEU of the restated scorer (pycocoevalcap's algorithm, transformer_faces_objects.py:109-116)."""
    from tell_amd.metrics import BleuScorer
    s = BleuScorer(n=4)
    s += ('the cat sat on the mat', ['the cat sat on the mat'])
    score, per = s.compute_score(option='closest')
    assert all(abs(x - 1.0) < 1e-6 for x in score)
    # hypothesis 'the the cat' vs reference 'the cat': unigrams clipped 2/3 (the:1 of 2, cat:1), bigrams 1/2 ('the cat'),
    # trigrams 0/1, 4-grams 0/0 -> tiny/small smoothing; longer than the reference: no brevity penalty
    s = BleuScorer(n=4)
    s += ('the the cat', ['the cat'])
    score, _ = s.compute_score(option='closest')
    assert abs(score[0] - 2 / 3) < 1e-6
    assert abs(score[1] - ((2 / 3) * (1 / 2)) ** 0.5) < 1e-6
    assert score[2] < 1e-4 and score[3] < 1e-2
    # brevity penalty: 2 of 4 words -> exp(1 - 4/2)
    s = BleuScorer(n=4)
    s += ('the cat', ['the cat sat down'])
    score, _ = s.compute_score(option='closest')
    import math
    assert abs(score[0] - math.exp(1 - 2.0)) < 1e-6
    # 'closest' reference length with two references
    s = BleuScorer(n=4)
    s += ('a b c', ['a b c d e f', 'a b'])
    assert s.ctest[0]['reflen'] == [6, 2] and s._single_reflen([6, 2], 'closest', 3) == 2


def test_rouge_l_and_cider_known_answers():
    """f4 (scripts/compute_metrics.py:146-148,178-179): ROUGE-L (beta 1.2) and CIDEr (n = 4, sigma = 6, clipped, Gaussian
    length penalty) as pycocoevalcap's scorers compute them - hand-computed cases."""
    import math
    from tell_amd.metrics import CiderScorer, Rouge
    # LCS("the cat sat on mat", "the cat is on the mat") = the cat on mat = 4: p = 4/5, r = 4/6
    p_, r_, b2 = 4 / 5, 4 / 6, 1.2 ** 2
    assert abs(Rouge().calc_score(['the cat sat on mat'], ['the cat is on the mat']) - (1 + b2) * p_ * r_ / (r_ + b2 * p_)) < 1e-12
    assert Rouge().calc_score(['a b c'], ['x y z']) == 0.0
    assert abs(Rouge().calc_score(['a b c'], ['a b c']) - 1.0) < 1e-12
    assert abs(Rouge().calc_score(['a b'], ['x', 'a b c']) - (1 + b2) * 1.0 * (2 / 3) / (2 / 3 + b2 * 1.0)) < 1e-12   # best ref per side
    # CIDEr: a hypothesis equal to its reference scores 10 once every order has an n-gram with non-zero idf
    c = CiderScorer(n=4, sigma=6.0)
    c += ('a b c d e', ['a b c d e'])
    c += ('f g h i j', ['k l m n o'])
    c += ('p q r s', ['p q r s t u'])
    corpus, per = c.compute_score()
    assert abs(per[0] - 10.0) < 1e-9 and per[1] == 0.0
    # third sample by hand: idf = ln 3 for every n-gram (each occurs in one reference set); hyp n-grams are a subset of
    # the reference's: cos_n = sqrt(#hyp n-grams / #ref n-grams) = sqrt((4-n)/(6-n)) for n = 0..3 (n+1 = order);
    # lengths are bigram counts 3 and 5: penalty exp(-4 / 72)
    want = 10.0 * math.exp(-4 / 72) * sum(math.sqrt((4 - k) / (6 - k)) for k in range(4)) / 4
    assert abs(per[2] - want) < 1e-9 and abs(corpus - (10.0 + want) / 3) < 1e-9
    # clipping: a repeated word cannot collect more than the reference holds
    d = CiderScorer(n=1, sigma=6.0)
    d += ('a a a a', ['a b'])
    d += ('x', ['y'])
    idf = math.log(2.0)
    vh, vr_a, vr_b = 4 * idf, idf, idf
    # (with n = 1 there are no bigrams: the scorer's 'length' is 0 on both sides, no penalty)
    want1 = 10.0 * (min(vh, vr_a) * vr_a) / (vh * math.sqrt(vr_a ** 2 + vr_b ** 2))
    assert abs(d.compute_score()[1][0] - want1) < 1e-9


def test_compute_metrics_over_a_generations_file(tmp_path):
    """The metric table of scripts/compute_metrics.py:179-298 over a generations.jsonl: text metrics always, name /
    entity / readability tables when the records carry the NLP-derived keys."""
    from collections import Counter
    from tell_amd.commands import compute_metrics
    recs = [
        {'raw_caption': 'Ann Lee, left, and Bob Ray in Paris.', 'generation': 'Ann Lee and Bob Ray in Paris',
         'caption_names': ['Ann Lee', 'Bob Ray'], 'generated_names': ['Ann Lee', 'Bob Ray'],
         'caption_entities': [{'text': 'Ann Lee', 'label': 'PERSON'}, {'text': 'Paris', 'label': 'GPE'}],
         'generated_entities': [{'text': 'Ann Lee', 'label': 'PERSON'}, {'text': 'Rome', 'label': 'GPE'}],
         'caption_np': {'basic_ttr': 1.0}, 'gen_np': {'basic_ttr': 0.5}},
        {'raw_caption': 'A quiet street in Rome.', 'generation': 'Cy Doe on a street',
         'caption_names': [], 'generated_names': ['Cy Doe'], 'caption_np': {'basic_ttr': 0.8}, 'gen_np': {'basic_ttr': 0.7}},
    ]
    path = tmp_path / 'generations.jsonl'
    path.write_text('\n'.join(json.dumps(r) for r in recs) + '\n')
    m = compute_metrics(str(path), counters={'caption': Counter({'Ann Lee': 3}), 'context': Counter({'Bob Ray': 1})})
    assert 0 < m['BLEU-4'] < m['BLEU-1'] <= 1 and 0 < m['ROUGE'] < 1 and m['CIDEr'] > 0 and m['METEOR'] is None
    assert m['All names - recall'] == {'count': 2, 'total': 2, 'percentage': 1.0}
    assert m['All names - precision'] == {'count': 2, 'total': 3, 'percentage': 2 / 3}
    assert m['Caption rare names - recall'] == {'count': 1, 'total': 1, 'percentage': 1.0}        # Bob Ray (Ann Lee is frequent)
    assert m['Article rare names - precision']['total'] == 1                                        # only Cy Doe is unseen
    assert m['Length - generation'] == (7 + 5) / 2 and m['Length - reference'] == (8 + 5) / 2
    assert m['Entity all - recall'] == {'count': 1, 'total': 2, 'percentage': 0.5}
    assert m['Entity GPE - precision'] == {'count': 0, 'total': 1, 'percentage': 0.0}
    assert abs(m['Generation TTR'] - 0.6) < 1e-12 and 'Generation Flesch Reading Ease' not in m


def test_shard_builder_restates_the_reference_reader_field_logic(tmp_path):
    """data/build_shards.py against nytimes_faces_ner_matched.py:104-190 on two hand-made article documents: one sample
    per image position; empty captions and unreadable images dropped; context = headline, FIRST paragraph, then the
    paragraphs around the image (document order before, then after) until 510 BPE tokens; PERSON names of the caption
    counted for the reader's face trimming; objects by image hash (missing document -> empty); and the shards it writes
    are what the reader serves."""
    from PIL import Image
    from tell_amd.data import DatasetReader, RobertaTokenIndexer
    from tell_amd.data.bpe import RobertaBPE
    from tell_amd.data.build_shards import article_samples, build_shards
    _write_vocab(str(tmp_path))
    idx = RobertaTokenIndexer(max_len=512, bpe=RobertaBPE(str(tmp_path)))
    img_dir = tmp_path / 'images'
    os.makedirs(img_dir)
    g = np.random.RandomState(0)
    pix = {}
    for h in ('h1', 'h2', 'h4'):
        pix[h] = g.randint(0, 256, (224, 224, 3)).astype(np.uint8)
        Image.fromarray(pix[h]).save(str(img_dir / ('%s.png' % h)))
        os.rename(str(img_dir / ('%s.png' % h)), str(img_dir / ('%s.jpg' % h)))          # lossless pixels under the .jpg name
    par = lambda t, ents=(): {'type': 'paragraph', 'text': t, 'named_entities': [{'text': e, 'label': l} for e, l in ents]}  # noqa: E731
    cap = lambda t, h, ents=(), faces=0: dict({'type': 'caption', 'text': t, 'hash': h,  # noqa: E731
                                               'named_entities': [{'text': e, 'label': l} for e, l in ents]},
                                              **({'facenet_details': {'embeddings': g.randn(faces, 512).tolist()}} if faces else {}))
    a1 = {'_id': 'a1', 'web_url': 'u1', 'headline': {'main': ' hello the '}, 'image_positions': [1, 4, 5, 6],
          'parsed_section': [par('p0', [('Milan', 'GPE')]), cap('the Milan', 'h1', [('Ann', 'PERSON'), ('Bo', 'PERSON'), ('X', 'ORG')], faces=3),
                             par('p2'), par('p3', [('Zed', 'PERSON')]), cap('hello', 'h2'), cap('   ', 'h3'), cap('the', 'missing')]}
    long_par = 'the ' * 300                                                                   # 300 tokens each
    a2 = {'_id': 'a2', 'web_url': 'u2', 'headline': {}, 'image_positions': [3],
          'parsed_section': [par('first'), par(long_par + 'A'), par(long_par + 'B'), cap('hello the', 'h4', faces=1),
                             par(long_par + 'C'), par(long_par + 'D')]}
    n_tok = lambda t: len(idx.bpe.bpe.encode(t))                                              # noqa: E731
    s1 = list(article_samples(a1, n_tok))
    assert [s[0] for s in s1] == [1, 4, 6]                                                    # position 5: empty caption
    pos, caption, paragraphs, named, section = s1[0]
    assert caption == 'the Milan' and paragraphs == ['hello the', 'p0', 'p2', 'p3'] and named == ['Milan', 'Zed']
    assert s1[1][2] == ['hello the', 'p0', 'p2', 'p3']                                        # before the image, document order
    s2 = list(article_samples(a2, n_tok))
    # first paragraph, then nearest-first around position 3: (B, C) = 600+ tokens >= 510 -> stop before A and D
    assert [p[-1] for p in s2[0][2]] == ['t', 'B', 'C'] and s2[0][2][0] == 'first'
    objects = {'h1': {'object_features': np.abs(g.randn(70, 2048)).tolist()}, 'h2': {'object_features': []}}
    written, skipped = build_shards([a2, a1], str(img_dir), str(tmp_path / 'shards'), 'train', idx, objects=objects,
                                    shard_size=2, shuffle_seed=None)
    assert (written, skipped) == (3, 1)                                                       # the 'missing' image is dropped
    assert sorted(os.listdir(tmp_path / 'shards')) == ['train-00000.npz', 'train-00001.npz']
    reader = DatasetReader.by_name('nytimes_faces_ner_matched')(shard_dir=str(tmp_path / 'shards'), use_objects=True,
                                                                use_caption_names=True, seed=0)
    got = {inst['metadata']['image_path'].split('/')[-1]: inst for inst in reader._read('train')}
    assert sorted(got) == ['h1.jpg', 'h2.jpg', 'h4.jpg']
    i1 = got['h1.jpg']
    assert np.array_equal(np.asarray(i1['image']), pix['h1'])
    assert i1['face_embeds'].shape == (2, 512)                                                # 3 faces stored, 2 PERSON names
    assert i1['obj_embeds'].shape == (64, 2048)                                               # capped at the model's 64
    assert list(i1['caption']['roberta'])[0] == 0 and list(i1['caption']['roberta'])[-1] == 2
    assert i1['metadata']['context'] == 'hello the\np0\np2\np3' and i1['metadata']['names'] == ['Milan', 'Zed']
    assert got['h2.jpg']['face_embeds'].shape == (1, 0) and got['h2.jpg']['obj_embeds'].shape[0] in (0, 1)
    assert got['h4.jpg']['obj_embeds'].shape[0] in (0, 1)                                     # no objects document: the empty array
    assert len(got['h4.jpg']['context']['roberta']) == 512                                    # indexer truncation (max_len)
