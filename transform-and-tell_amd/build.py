"""Programmatic constructors equivalent to the `model:` section of the expt/ configs
(used by tests, bench.py and smoke(); `config.py` builds the same objects from YAML)."""
from .models.decoders import (DynamicConvDecoder, DynamicConvDecoderNoImage, DynamicConvFacesObjectsDecoder,
                              DynamicConvFacesParallelDecoder)
from .models.transformer import TransformerFacesModel, TransformerFacesObjectModel, TransformerFlattenedModel
from .modules import AdaptiveEmbedding, AdaptiveLoss, SinusoidalPositionalEmbedding, SumTextFieldEmbedder


def build_embedder(vocab_size=50265, dim=1024, cutoff=(5000, 20000), init_size=512):
    return SumTextFieldEmbedder(
        {'adaptive': AdaptiveEmbedding(None, 'bpe', 0, dim, 1, dim, list(cutoff), vocab_size=vocab_size,
                                       scale_embeds=True),
         'position': SinusoidalPositionalEmbedding(None, dim, 1, False, init_size=init_size)},
        embedder_to_indexer_map={'adaptive': ['roberta'], 'position': ['roberta']}, allow_unmatched_keys=True)


def decoder_kwargs(vocab_size=50265, dim=1024, heads=16, ffn=4096, kernels=(3, 7, 15, 31), cutoff=(5000, 20000)):
    """expt/nytimes/9_transformer_objects/config.yaml:51-76"""
    return dict(max_target_positions=512, dropout=0.1, share_decoder_input_output_embed=True,
                decoder_output_dim=dim, decoder_conv_dim=dim, decoder_glu=True, decoder_conv_type='dynamic',
                weight_softmax=True, decoder_attention_heads=heads, weight_dropout=0.1, relu_dropout=0.0,
                input_dropout=0.1, decoder_normalize_before=False, attention_dropout=0.1,
                decoder_ffn_embed_dim=ffn, decoder_kernel_size_list=list(kernels),
                adaptive_softmax_cutoff=list(cutoff), adaptive_softmax_factor=1, tie_adaptive_weights=True,
                adaptive_softmax_dropout=0, tie_adaptive_proj=False, decoder_layers=len(kernels),
                final_norm=False, padding_idx=0, namespace='bpe', vocab_size=vocab_size)


def build_decoder(kind='faces_objects', vocab_size=50265, dim=1024, heads=16, ffn=4096, kernels=(3, 7, 15, 31),
                  cutoff=(5000, 20000), article_dim=1024, init_size=512):
    emb = build_embedder(vocab_size, dim, cutoff, init_size)
    kw = decoder_kwargs(vocab_size, dim, heads, ffn, kernels, cutoff)
    if kind.endswith('_prenorm'):                            # pre-LN blocks + final LayerNorm, no GLU
        kw.update(decoder_normalize_before=True, final_norm=True, decoder_glu=False)
    if kind.endswith('_lightweight'):                        # `decoder_conv_type: lightweight` (decoder_flattened.py:199-203)
        kw['decoder_conv_type'] = 'lightweight'
    if kind == 'faces_objects':
        return DynamicConvFacesObjectsDecoder(None, emb, **kw)
    if kind in ('faces_parallel', 'faces'):                  # expt/*/8_transformer_faces
        return DynamicConvFacesParallelDecoder(None, emb, **kw)
    if kind == 'flattened_no_image':                         # expt/*/4_no_image
        return DynamicConvDecoderNoImage(None, emb, article_embed_size=article_dim, **kw)
    return DynamicConvDecoder(None, emb, article_embed_size=article_dim, **kw)


def build_model(kind, resnet=None, roberta=None, weigh_bert=True, n_bert_layers=25, **decoder_kw):
    dec = build_decoder(kind, **decoder_kw)
    cls = {'faces_objects': TransformerFacesObjectModel, 'faces': TransformerFacesModel,
           'faces_parallel': TransformerFacesModel}.get(kind, TransformerFlattenedModel)
    return cls(None, dec, AdaptiveLoss(padding_idx=1), weigh_bert=weigh_bert, vocab_size=decoder_kw.get('vocab_size', 50265),
               resnet=resnet, roberta=roberta, n_bert_layers=n_bert_layers)
