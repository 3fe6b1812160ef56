"""YAML experiment configs -> objects (the `from_params` side of the plugin surface).

The reference turns `expt/**/config.yaml` into AllenNLP `Params` (tell/commands/train.py:67-77)
and lets `Registrable.from_params` build Model / Decoder / Criterion / TextFieldEmbedder /
TokenEmbedder / DatasetReader / Trainer from the `type:` keys.  AllenNLP is not installable
here, so this is the small equivalent for the hot path's types: same registration strings,
constructor kwargs == YAML keys."""
import copy
import json

import yaml

from .data.readers import DatasetReader
from .data.indexers import Vocabulary
from .data.iterators import DataIterator
from .models.decoders import Decoder
from .models.transformer import Model
from .modules.criteria import Criterion
from .modules.token_embedders import TextFieldEmbedder, TokenEmbedder
from .training.trainer import TrainerBase


def _merge(base, over):
    for k, v in over.items():
        if isinstance(v, dict) and isinstance(base.get(k), dict):
            _merge(base[k], v)
        else:
            base[k] = v
    return base


def yaml_to_params(path, overrides=''):
    """tell/commands/train.py:67-77: yaml.safe_load + JSON overrides."""
    with open(path) as f:
        params = yaml.safe_load(f)
    if overrides:
        _merge(params, json.loads(overrides))
    return params


def embedder_from_params(p, vocab=None):
    p = copy.deepcopy(p)
    cls = TextFieldEmbedder.by_name(p.pop('type'))
    tok = {}
    for name, sub in p.pop('token_embedders').items():
        sub = dict(sub)
        tcls = TokenEmbedder.by_name(sub.pop('type'))
        tok[name] = tcls(vocab, **sub)
    return cls(tok, p.pop('embedder_to_indexer_map', None), p.pop('allow_unmatched_keys', False))


def decoder_from_params(p, vocab=None):
    p = copy.deepcopy(p)
    cls = Decoder.by_name(p.pop('type'))
    embedder = embedder_from_params(p.pop('embedder'), vocab)
    return cls(vocab, embedder, **p)


def model_from_params(p, vocab=None, **extra):
    """`model:` section of a config -> Model (weights random-initialised; load a checkpoint with
    model.load_state_dict(torch.load(path)) exactly as tell/commands/evaluate.py:61-63 does)."""
    p = copy.deepcopy(p)
    cls = Model.by_name(p.pop('type'))
    decoder = decoder_from_params(p.pop('decoder'), vocab)
    cp = dict(p.pop('criterion'))
    criterion = Criterion.by_name(cp.pop('type'))(**cp)
    p.pop('initializer', None)
    p.update(extra)
    return cls(vocab, decoder, criterion, **p)


def reader_from_params(p, **extra):
    p = dict(p)
    cls = DatasetReader.by_name(p.pop('type'))
    p.update(extra)
    return cls(**p)


def vocabulary_from_params(p):
    """`vocabulary:` section (config.yaml:22-24: `type: roberta`, directory_path)."""
    p = dict(p)
    return Vocabulary.by_name(p.pop('type'))(**p)


def iterator_from_params(p):
    """`iterator:` / `validation_iterator:` sections (config.yaml:99-114: `type: bucket`)."""
    p = dict(p)
    return DataIterator.by_name(p.pop('type'))(**p)


def trainer_from_params(p, model):
    p = copy.deepcopy(p)
    cls = TrainerBase.by_name(p.pop('type'))
    return cls(model, **p)


def from_config(path, overrides='', **model_extra):
    params = yaml_to_params(path, overrides)
    return model_from_params(params['model'], **model_extra), params
