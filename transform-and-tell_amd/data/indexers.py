"""tell/data/token_indexers/roberta_indexer.py:34-200 and tell/data/vocabulary.py:29 on the MI355X path's data plane:
same registration names (`TokenIndexer: roberta`, `Vocabulary: roberta`), same constructor keys (config.yaml:6-13,
:22-24), same tensors out (`<name>` int64 padded with 1, `<name>_copy_masks` padded with -1)."""
import os

import torch

from ..common.registrable import Registrable
from .bpe import RobertaBPE


class TokenIndexer(Registrable):
    pass


class Vocabulary(Registrable):
    pass


def bpe_directory(model_name='roberta-base'):
    """Where the three BPE files live: $TELL_BPE_DIR, else ~/.cache/tell_amd/<model_name> (no download: no network)."""
    return os.environ.get('TELL_BPE_DIR') or os.path.join(os.path.expanduser('~'), '.cache', 'tell_amd', model_name)


@TokenIndexer.register('roberta')
class RobertaTokenIndexer(TokenIndexer):
    def __init__(self, model_name='roberta-base', namespace='bpe', legacy=False, start_tokens=None, end_tokens=None,
                 token_min_padding_length=0, padding_on_right=True, padding_value=1, max_len=512, bpe=None):
        """bpe: a ready RobertaBPE (tests); otherwise the files are looked up lazily at the first `encode`, so that a
        config instantiates on a box without them (pre-indexed shards never call `encode`)."""
        if not padding_on_right:
            raise NotImplementedError('right padding only (padding_on_right: true in every config)')
        self.model_name, self._namespace = model_name, namespace
        self._padding_value, self._max_len, self.legacy = padding_value, max_len, legacy
        self._bpe = bpe

    @property
    def bpe(self):
        if self._bpe is None:
            self._bpe = RobertaBPE(bpe_directory(self.model_name))
        return self._bpe

    # ---- roberta_indexer.py:89-183
    def encode(self, sentence, doc=None):
        """-> (token ids with <s> / </s>, copy masks); doc: optional spaCy Doc (entity spans -> copy mask 1)."""
        rb = self.bpe
        raw = rb.bpe.pretokenize(sentence)
        masks = self.get_entity_mask(raw, doc)
        ids, copy = [], []
        for tok, m in zip(raw, masks):
            piece = rb.bpe.encode_pretoken(tok)
            ids.extend(piece)
            copy.extend([1 if m else 0] * len(piece))
        ids, copy = ids[:self._max_len - 2], copy[:self._max_len - 2]
        d = rb.source_dictionary
        token_ids = [d.bos_index] + [d.indices[str(i)] for i in ids] + [d.eos_index]      # KeyError like the reference
        return token_ids, [0] + copy + [0]

    @staticmethod
    def get_entity_mask(tokens, doc):
        starts, ends, cur = [], [], 0
        for t in tokens:
            starts.append(cur)
            cur += len(t)
            ends.append(cur)
        masks = [0] * len(tokens)
        if doc is None:
            return masks
        for ent in doc.ents:
            for i, (s, e, t) in enumerate(zip(starts, ends, tokens)):
                es = ent.start_char - (1 if t[0] == ' ' else 0)
                if s >= es and e <= ent.end_char:
                    masks[i] = 1
        return masks

    def tokens_to_indices(self, tokens, vocabulary=None, index_name='roberta', doc=None):
        """tokens: list of strings (or objects with .text), joined by spaces as the reference does (:78)."""
        text = ' '.join(getattr(t, 'text', t) for t in tokens)
        ids, copy = self.encode(text, doc)
        return {index_name: ids, index_name + '_copy_masks': copy}

    # ---- roberta_indexer.py:185-200
    def as_padded_tensor(self, tokens, desired_num_tokens, padding_lengths=None):
        out = {}
        for key, val in tokens.items():
            pad = -1 if 'copy_masks' in key else self._padding_value
            n = desired_num_tokens[key]
            out[key] = torch.tensor((list(val) + [pad] * n)[:n], dtype=torch.long)
        return out


@Vocabulary.register('roberta')
class RobertaVocabulary(Vocabulary):
    """tell/data/vocabulary.py:29-100: a vocabulary whose pad / unk indices are RoBERTa's (1 / 3); the BPE namespace is
    filled by the indexer.  Only what the hot path reads is kept: `get_vocab_size` and the two special indices."""

    def __init__(self, directory_path=None, padding_token='<pad>', oov_token='<unk>', **unused):
        self.directory_path = directory_path
        self._token_to_index = {'bpe': {padding_token: 1, oov_token: 3}}
        self._index_to_token = {'bpe': {1: padding_token, 3: oov_token}}

    @classmethod
    def from_files(cls, directory):
        return cls(directory_path=directory)

    def add_indexer(self, indexer):
        for piece, idx in indexer.bpe.source_dictionary.indices.items():
            self._token_to_index['bpe'][piece] = idx
            self._index_to_token['bpe'][idx] = piece

    def get_vocab_size(self, namespace='bpe'):
        return max(len(self._token_to_index.get(namespace, {})), 0)
