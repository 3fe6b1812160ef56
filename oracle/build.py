"""Constructors that assemble oracle modules the way the expt/ configs do.

TEST INFRASTRUCTURE - see oracle/__init__.py.
"""
from .decoders import (CONTEXTS_FACES_OBJECTS, CONTEXTS_FACES_PARALLEL, CONTEXTS_FLATTENED, CONTEXTS_NO_IMAGE,
                       DynamicConvDecoder)
from .models import CaptionModel
from .modules import (AdaptiveEmbedding, AdaptiveLoss, SinusoidalPositionalEmbedding,
                      SumTextFieldEmbedder)


def build_embedder(vocab_size=50265, dim=1024, cutoff=(5000, 20000), init_size=512):
    """expt/nytimes/9_transformer_objects/config.yaml:28-50."""
    return SumTextFieldEmbedder(
        {'adaptive': AdaptiveEmbedding(vocab_size, 0, dim, 1, dim, list(cutoff), scale_embeds=True),
         'position': SinusoidalPositionalEmbedding(dim, 1, False, init_size=init_size)},
        embedder_to_indexer_map={'adaptive': ['roberta'], 'position': ['roberta']},
        allow_unmatched_keys=True)


def build_decoder(kind='faces_objects', vocab_size=50265, dim=1024, heads=16, ffn=4096,
                  kernels=(3, 7, 15, 31), cutoff=(5000, 20000), article_dim=1024, **overrides):
    """config.yaml:26-76 (`dynamic_conv_decoder_faces_objects`) or
    expt/nytimes/5_transformer_roberta (`dynamic_conv_decoder_flattened`)."""
    if kind == 'faces_objects':
        contexts = CONTEXTS_FACES_OBJECTS
    elif kind == 'faces_parallel':                       # expt/*/8_transformer_faces
        contexts = CONTEXTS_FACES_PARALLEL
    elif kind == 'flattened_no_image':                   # expt/*/4_no_image
        contexts = (('article', article_dim),)
    else:                                                # 'flattened', 'flattened_lightweight'
        contexts = (CONTEXTS_FLATTENED[0], ('article', article_dim))
    if kind.endswith('_prenorm'):                        # decoder_normalize_before + final_norm, decoder_glu: false
        overrides.setdefault('decoder_normalize_before', True)
        overrides.setdefault('final_norm', True)
        overrides.setdefault('decoder_glu', False)
    if kind.endswith('_lightweight'):                    # `decoder_conv_type: lightweight` (decoder_flattened.py:199-203)
        overrides.setdefault('decoder_conv_type', 'lightweight')
    kw = dict(decoder_conv_dim=dim, decoder_attention_heads=heads, decoder_ffn_embed_dim=ffn,
              decoder_kernel_size_list=tuple(kernels), adaptive_softmax_cutoff=tuple(cutoff),
              decoder_layers=len(kernels), vocab_size=vocab_size)
    kw.update(overrides)
    return DynamicConvDecoder(build_embedder(vocab_size, dim, cutoff), contexts, **kw)


def build_model(kind, resnet, roberta, n_bert_layers=25, **decoder_kw):
    dec = build_decoder(kind, **decoder_kw)
    return CaptionModel(dec, AdaptiveLoss(padding_idx=1), resnet, roberta,
                        use_faces_objects=(kind in ('faces_objects', 'faces_parallel')), weigh_bert=True,
                        n_bert_layers=n_bert_layers)
