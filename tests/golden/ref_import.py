"""Import the *real* reference (`/root/reference/tell`) in the authoring container.

Test infrastructure only.  Used by `make_golden.py` to produce the committed
fixtures (`*.npz`); never imported by `-m gpu` tests, `smoke()` or `bench.py`
(the reference does not exist on the GPU box).

The reference's package `__init__`s import AllenNLP (absent here, not
installable).  We therefore
  1. register `tell`, `tell.models`, `tell.modules` ... as bare namespace
     modules whose `__path__` points into /root/reference so that the package
     `__init__.py` files never execute, and
  2. provide ~40 lines of stand-ins for the handful of AllenNLP / overrides /
     pycocoevalcap / torchvision symbols the hot-path files import at module
     import time (base classes and decorators only - no arithmetic).
No reference source is copied; the modules are executed from where they lie.
"""
import importlib
import os
import sys
import types

import torch
import torch.nn as nn

REF_ROOT = os.environ.get('TELL_REFERENCE_ROOT', '/root/reference')


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _Registrable:
    _registry = {}

    @classmethod
    def register(cls, name):
        def deco(sub):
            _Registrable._registry.setdefault(cls.__name__, {})[name] = sub
            return sub
        return deco


class _TokenEmbedder(nn.Module, _Registrable):
    def get_output_dim(self):
        raise NotImplementedError


class _TextFieldEmbedder(nn.Module, _Registrable):
    pass


class _Model(nn.Module, _Registrable):
    def __init__(self, vocab=None, regularizer=None):
        super().__init__()
        self.vocab = vocab


class _InitializerApplicator:
    def __init__(self, *a, **k):
        pass

    def __call__(self, module):
        return None


class _BleuScorer:
    def __init__(self, n=4):
        self.n = n

    def __iadd__(self, other):
        return self

    def compute_score(self, option=None):
        return [0.0] * self.n, None


def install_stubs():
    if 'allennlp' in sys.modules and getattr(sys.modules['allennlp'], '_tell_stub', False):
        return

    def overrides(f):
        return f
    _mod('overrides', overrides=overrides)

    class Params(dict):
        pass

    class ConfigurationError(Exception):
        pass

    class Vocabulary:
        pass

    class TimeDistributed(nn.Module):
        def __init__(self, m):
            super().__init__()
            self._module = m

    a = _mod('allennlp', _tell_stub=True)
    _mod('allennlp.common', Params=Params, Registrable=_Registrable)
    _mod('allennlp.common.registrable', Registrable=_Registrable)
    _mod('allennlp.common.checks', ConfigurationError=ConfigurationError)
    _mod('allennlp.data', Vocabulary=Vocabulary)
    _mod('allennlp.data.vocabulary', Vocabulary=Vocabulary)
    _mod('allennlp.models', Model=_Model)
    _mod('allennlp.models.model', Model=_Model)
    _mod('allennlp.nn')
    _mod('allennlp.nn.initializers', InitializerApplicator=_InitializerApplicator)
    _mod('allennlp.modules')
    _mod('allennlp.modules.token_embedders', TokenEmbedder=_TokenEmbedder)
    _mod('allennlp.modules.token_embedders.token_embedder', TokenEmbedder=_TokenEmbedder)
    _mod('allennlp.modules.text_field_embedders', TextFieldEmbedder=_TextFieldEmbedder)
    _mod('allennlp.modules.text_field_embedders.text_field_embedder',
         TextFieldEmbedder=_TextFieldEmbedder)
    _mod('allennlp.modules.time_distributed', TimeDistributed=TimeDistributed)
    _mod('pycocoevalcap')
    _mod('pycocoevalcap.bleu')
    _mod('pycocoevalcap.bleu.bleu_scorer', BleuScorer=_BleuScorer)

    # namespace packages: package __init__ files are NOT executed
    for pkg in ['tell', 'tell.models', 'tell.modules', 'tell.modules.attention',
                'tell.modules.convolutions', 'tell.modules.criteria',
                'tell.modules.token_embedders']:
        m = types.ModuleType(pkg)
        m.__path__ = [os.path.join(REF_ROOT, *pkg.split('.'))]
        sys.modules[pkg] = m
    # tell.utils has a clean __init__ (pure python) -> import for real
    importlib.import_module('tell.utils')

    # the decoders do `from tell.modules import (...)`; populate lazily
    tm = sys.modules['tell.modules']
    from tell.modules.convolutions.dynamic import DynamicConv1dTBC
    from tell.modules.convolutions.lightweight import LightweightConv1dTBC
    from tell.modules.attention.multi_head import MultiHeadAttention
    from tell.modules.linear import GehringLinear
    from tell.modules.softmax import AdaptiveSoftmax
    tm.DynamicConv1dTBC = DynamicConv1dTBC
    tm.LightweightConv1dTBC = LightweightConv1dTBC
    tm.MultiHeadAttention = MultiHeadAttention
    tm.GehringLinear = GehringLinear
    tm.AdaptiveSoftmax = AdaptiveSoftmax
    te = sys.modules['tell.modules.token_embedders']
    from tell.modules.token_embedders.adaptive import AdaptiveEmbedding
    from tell.modules.token_embedders.positional import SinusoidalPositionalEmbedding
    from tell.modules.token_embedders.sum_text_field_embedder import SumTextFieldEmbedder
    te.AdaptiveEmbedding = AdaptiveEmbedding
    te.SinusoidalPositionalEmbedding = SinusoidalPositionalEmbedding
    te.SumTextFieldEmbedder = SumTextFieldEmbedder
    tc = sys.modules['tell.modules.criteria']
    from tell.modules.criteria.base import Criterion
    from tell.modules.criteria.adaptive_loss import AdaptiveLoss
    tc.Criterion = Criterion
    tc.AdaptiveLoss = AdaptiveLoss


class StandInEncoders:
    """Deterministic stand-ins for the two pretrained encoders whose code is not
    under /root/reference (fairseq RoBERTa via torch.hub; torchvision ResNet).
    They only provide tensors of the right shape so that the reference's own
    `_forward` / `forward` / `_generate` code runs unmodified."""

    class Roberta(nn.Module):
        def __init__(self, n_layers=25, dim=1024, vocab=50265, seed=0):
            super().__init__()
            g = torch.Generator().manual_seed(seed)
            self.tables = nn.Parameter(torch.randn(n_layers, 64, dim, generator=g) * 0.5,
                                       requires_grad=False)
            self.n_layers = n_layers

        def extract_features(self, ids, return_all_hiddens=False):
            outs = [self.tables[i][ids % 64] for i in range(self.n_layers)]
            return outs if return_all_hiddens else outs[-1]

        def decode(self, ids):
            return ' '.join(str(int(i)) for i in ids)

    class Resnet(nn.Module):
        def __init__(self, seed=0):
            super().__init__()
            g = torch.Generator().manual_seed(seed)
            self.proj = nn.Parameter(torch.randn(2048, 3, generator=g) * 0.3,
                                     requires_grad=False)

        def forward(self, image):
            # [B,3,224,224] -> [B,2048,7,7]: 32x32 mean pool then 3->2048 mix
            p = torch.nn.functional.avg_pool2d(image, 32)
            return torch.relu(torch.einsum('oc,bchw->bohw', self.proj, p))


def import_models():
    """Import the reference model files with the stand-in encoders wired in."""
    install_stubs()
    # tell.models.resnet imports torchvision (absent) -> pre-register a stand-in module
    _mod('tell.models.resnet', resnet152=lambda *a, **k: StandInEncoders.Resnet())
    _orig_hub = torch.hub.load
    torch.hub.load = lambda *a, **k: StandInEncoders.Roberta()
    try:
        from tell.models import decoder_base  # noqa
        dfo = importlib.import_module('tell.models.decoder_faces_objects')
        dfl = importlib.import_module('tell.models.decoder_flattened')
        tfo = importlib.import_module('tell.models.transformer_faces_objects')
        tfl = importlib.import_module('tell.models.transformer_flattened')
    finally:
        pass
    return dfo, dfl, tfo, tfl, _orig_hub


def import_decoder_variants():
    """The context-table siblings of the two decoders: 3-context `faces_parallel`
    (expt/*/8_transformer_faces, a1-a3) and article-only `flattened_no_image` (expt/*/4_no_image)."""
    import_models()
    return (importlib.import_module('tell.models.decoder_faces_parallel'),
            importlib.import_module('tell.models.decoder_flattened_no_image'))
