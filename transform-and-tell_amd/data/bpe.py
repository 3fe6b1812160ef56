"""Byte-level BPE of RoBERTa (GPT-2's tokenizer) + the fairseq dictionary on top of it - what
`torch.hub.load('pytorch/fairseq:2f7e3f3323', 'roberta.base')` gives the reference's indexer
(tell/data/token_indexers/roberta_indexer.py:46-48: `roberta.bpe.bpe`, `roberta.task.source_dictionary`).

fairseq is absent here (third-party, pinned by git commit in the reference); this is a restatement of the published
GPT-2 encoder algorithm and of fairseq's Dictionary file format:
  encoder.json  {bpe token string -> GPT-2 id}          vocab.bpe  ranked merges ("a b" per line, first line a header)
  dict.txt      one "<GPT-2 id as text> <count>" per line; fairseq index = 4 + line number after the specials
                <s>=0 <pad>=1 </s>=2 <unk>=3
The three files are looked up in `directory` (no network here: tests write a small synthetic vocabulary)."""
import json
import os
from functools import lru_cache

import regex


@lru_cache()
def bytes_to_unicode():
    """The reversible byte -> printable-unicode map of GPT-2 (spaces / control bytes get code points >= 256)."""
    bs = list(range(ord('!'), ord('~') + 1)) + list(range(ord('\xa1'), ord('\xac') + 1)) + \
        list(range(ord('\xae'), ord('\xff') + 1))
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    return dict(zip(bs, [chr(c) for c in cs]))


PATTERN = regex.compile(r"""'s|'t|'re|'ve|'m|'ll|'d| ?\p{L}+| ?\p{N}+| ?[^\s\p{L}\p{N}]+|\s+(?!\S)|\s+""")


class ByteBPE:
    def __init__(self, encoder, merges):
        self.encoder = dict(encoder)
        self.decoder = {v: k for k, v in self.encoder.items()}
        self.bpe_ranks = {tuple(m): i for i, m in enumerate(merges)}
        self.byte_encoder = bytes_to_unicode()
        self.byte_decoder = {v: k for k, v in self.byte_encoder.items()}
        self.pat = PATTERN
        self.cache = {}

    @classmethod
    def from_files(cls, encoder_json, vocab_bpe):
        with open(encoder_json) as f:
            encoder = json.load(f)
        with open(vocab_bpe, encoding='utf-8') as f:
            lines = f.read().split('\n')
        merges = [tuple(ln.split()) for ln in lines[1:] if ln.strip()]
        return cls(encoder, merges)

    def bpe(self, token):
        """Merge the symbols of one pre-token by rank; -> space-joined BPE tokens."""
        if token in self.cache:
            return self.cache[token]
        word = tuple(token)
        while len(word) > 1:
            pairs = {(word[i], word[i + 1]) for i in range(len(word) - 1)}
            best = min(pairs, key=lambda p: self.bpe_ranks.get(p, float('inf')))
            if best not in self.bpe_ranks:
                break
            first, second = best
            new, i = [], 0
            while i < len(word):
                if i < len(word) - 1 and word[i] == first and word[i + 1] == second:
                    new.append(first + second)
                    i += 2
                else:
                    new.append(word[i])
                    i += 1
            word = tuple(new)
        out = ' '.join(word)
        self.cache[token] = out
        return out

    def pretokenize(self, text):
        return self.pat.findall(text)

    def encode_pretoken(self, raw_token):
        token = ''.join(self.byte_encoder[b] for b in raw_token.encode('utf-8'))
        return [self.encoder[t] for t in self.bpe(token).split(' ')]

    def encode(self, text):
        ids = []
        for raw in self.pretokenize(text):
            ids.extend(self.encode_pretoken(raw))
        return ids

    def decode(self, ids):
        text = ''.join(self.decoder[int(i)] for i in ids)
        return bytearray(self.byte_decoder[c] for c in text).decode('utf-8', errors='replace')


class FairseqDictionary:
    """fairseq Dictionary restricted to what the hot path uses: symbol <-> index."""

    def __init__(self, symbols):
        self.symbols = ['<s>', '<pad>', '</s>', '<unk>'] + list(symbols)
        self.indices = {s: i for i, s in enumerate(self.symbols)}
        self.bos_index, self.pad_index, self.eos_index, self.unk_index = 0, 1, 2, 3

    @classmethod
    def load(cls, path):
        with open(path, encoding='utf-8') as f:
            return cls([ln.rsplit(' ', 1)[0] for ln in f.read().split('\n') if ln.strip()])

    def __len__(self):
        return len(self.symbols)


class RobertaBPE:
    """`roberta.bpe` + `roberta.task.source_dictionary` of the reference in one object."""

    FILES = ('encoder.json', 'vocab.bpe', 'dict.txt')

    def __init__(self, directory):
        missing = [f for f in self.FILES if not os.path.exists(os.path.join(directory, f))]
        if missing:
            raise FileNotFoundError('RoBERTa BPE files %s not found in %r (fairseq downloads them with the model; copy '
                                    'encoder.json, vocab.bpe and dict.txt of roberta.base there)' % (missing, directory))
        self.bpe = ByteBPE.from_files(os.path.join(directory, 'encoder.json'), os.path.join(directory, 'vocab.bpe'))
        self.source_dictionary = FairseqDictionary.load(os.path.join(directory, 'dict.txt'))

    def encode_ids(self, text, max_len=512):
        """text -> fairseq ids with <s> ... </s>, truncated to max_len (roberta_indexer.py:89-109)."""
        words = [str(i) for i in self.bpe.encode(text)][:max_len - 2]
        d = self.source_dictionary
        return [d.bos_index] + [d.indices.get(w, d.unk_index) for w in words] + [d.eos_index]

    def decode(self, ids):
        """fairseq `roberta.decode` for a 1-D id sequence without <s>/<pad> (transformer_faces_objects.py:96):
        stops at the first </s>, unknown ids are dropped."""
        d = self.source_dictionary
        out = []
        for i in [int(x) for x in ids]:
            if i == d.eos_index:
                break
            if i < 4 or i >= len(d):
                continue
            sym = d.symbols[i]
            if sym.isdigit():
                out.append(int(sym))
        return self.bpe.decode(out)
