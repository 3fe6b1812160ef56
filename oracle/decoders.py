"""Table-driven DynamicConv decoder (2- and 4-context) for the oracle.

TEST INFRASTRUCTURE - see oracle/__init__.py.

Restates tell/models/decoder_faces_objects.py:22-379 and
tell/models/decoder_flattened.py:23-333: the two files are the same block
structure over a different list of contexts, so the oracle has one class
parameterised by `contexts = [(name, kdim), ...]`.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .modules import (AdaptiveSoftmax, DynamicConv1dTBC, GehringLinear, LightweightConv1dTBC, MultiHeadAttention,
                      _maybe_dropout)

CONTEXTS_FLATTENED = (('image', 2048), ('article', 1024))
CONTEXTS_FACES_OBJECTS = (('image', 2048), ('article', 1024), ('faces', 512), ('obj', 2048))
CONTEXTS_FACES_PARALLEL = (('image', 2048), ('article', 1024), ('faces', 512))    # decoder_faces_parallel.py:22
CONTEXTS_NO_IMAGE = (('article', 1024),)                                         # decoder_flattened_no_image.py:22


class DynamicConvDecoderLayer(nn.Module):
    """decoder_faces_objects.py:184-372 / decoder_flattened.py:185-326."""

    def __init__(self, embed_dim, conv_dim, glu, heads, weight_dropout, dropout, relu_dropout,
                 input_dropout, normalize_before, attention_dropout, ffn_dim, kernel_size, contexts,
                 conv_type='dynamic'):
        super().__init__()
        self.linear1 = GehringLinear(embed_dim, 2 * conv_dim if glu else conv_dim)
        self.glu = glu
        conv_cls = {'dynamic': DynamicConv1dTBC, 'lightweight': LightweightConv1dTBC}[conv_type]   # :199-211
        self.conv = conv_cls(conv_dim, kernel_size, heads, weight_dropout)
        self.linear2 = GehringLinear(conv_dim, embed_dim)
        self.dropout, self.relu_dropout, self.input_dropout = dropout, relu_dropout, input_dropout
        self.normalize_before = normalize_before
        self.conv_layer_norm = nn.LayerNorm(embed_dim)
        self.context_attns = nn.ModuleDict()
        self.context_attn_lns = nn.ModuleDict()
        self.context_names = [n for n, _ in contexts]
        for name, kdim in contexts:
            self.context_attns[name] = MultiHeadAttention(embed_dim, heads, kdim=kdim, vdim=kdim,
                                                          dropout=attention_dropout)
            self.context_attn_lns[name] = nn.LayerNorm(embed_dim)
        self.context_fc = GehringLinear(embed_dim * len(contexts), embed_dim)
        self.fc1 = GehringLinear(embed_dim, ffn_dim)
        self.fc2 = GehringLinear(ffn_dim, embed_dim)
        self.final_layer_norm = nn.LayerNorm(embed_dim)
        self.need_attn = True

    def _ln(self, ln, x, before):
        # maybe_layer_norm (decoder_faces_objects.py:367-372)
        return ln(x) if (before == self.normalize_before) else x

    def forward(self, x, contexts, incremental_state, need_weights=None):
        tr = self.training
        res = x                                                        # :256-266 conv block
        x = self._ln(self.conv_layer_norm, x, True)
        x = _maybe_dropout(x, self.input_dropout, tr)
        x = self.linear1(x)
        if self.glu:
            x = F.glu(x, dim=-1)
        x = self.conv(x, incremental_state=incremental_state)
        x = self.linear2(x)
        x = res + _maybe_dropout(x, self.dropout, tr)
        x = self._ln(self.conv_layer_norm, x, False)

        if need_weights is None:
            need_weights = (not tr) and self.need_attn
        attns, outs = {}, []
        for name in self.context_names:                                # :271-352
            h = self._ln(self.context_attn_lns[name], x, True)
            h, w = self.context_attns[name](h, contexts[name], contexts[name + '_mask'],
                                            need_weights=need_weights)
            h = x + _maybe_dropout(h, self.dropout, tr)
            outs.append(self._ln(self.context_attn_lns[name], h, False))
            if w is not None:
                attns[name] = w.detach().float().numpy()
        x = self.context_fc(torch.cat(outs, dim=-1))                   # :354-355 (no residual)

        res = x                                                        # :357-364 FFN
        x = self._ln(self.final_layer_norm, x, True)
        x = _maybe_dropout(F.relu(self.fc1(x)), self.relu_dropout, tr)
        x = res + _maybe_dropout(self.fc2(x), self.dropout, tr)
        return self._ln(self.final_layer_norm, x, False), attns


class DynamicConvDecoder(nn.Module):
    """decoder_faces_objects.py:22-180 / decoder_flattened.py:23-181 with an
    adaptive-softmax head tied to the adaptive input embedding."""

    def __init__(self, embedder, contexts, dropout=0.1, decoder_conv_dim=1024, decoder_glu=True,
                 decoder_attention_heads=16, weight_dropout=0.1, relu_dropout=0.0,
                 input_dropout=0.1, decoder_normalize_before=False, attention_dropout=0.1,
                 decoder_ffn_embed_dim=4096, decoder_kernel_size_list=(3, 7, 15, 31),
                 adaptive_softmax_cutoff=(5000, 20000), decoder_layers=4, final_norm=False,
                 vocab_size=50265, max_target_positions=512, decoder_conv_type='dynamic'):
        super().__init__()
        self.embedder = embedder
        E = embedder.get_output_dim()
        self.dropout = dropout
        self.max_target_positions = max_target_positions
        self.layers = nn.ModuleList([
            DynamicConvDecoderLayer(E, decoder_conv_dim, decoder_glu, decoder_attention_heads,
                                    weight_dropout, dropout, relu_dropout, input_dropout,
                                    decoder_normalize_before, attention_dropout,
                                    decoder_ffn_embed_dim, decoder_kernel_size_list[i], contexts,
                                    conv_type=decoder_conv_type)
            for i in range(decoder_layers)])
        self.adaptive_softmax = AdaptiveSoftmax(vocab_size, E, list(adaptive_softmax_cutoff),
                                                embedder.token_embedder_adaptive)
        self.register_buffer('version', torch.Tensor([2]))
        self.normalize = decoder_normalize_before and final_norm
        if self.normalize:
            self.layer_norm = nn.LayerNorm(E)

    def forward(self, prev_target, contexts, incremental_state=None, need_weights=None):
        x = self.embedder(prev_target, incremental_state=incremental_state)   # :98
        x = _maybe_dropout(x, self.dropout, self.training).transpose(0, 1)    # :106-109
        attns, inner = [], [x]
        for layer in self.layers:
            x, a = layer(x, contexts, incremental_state, need_weights)
            inner.append(x)
            attns.append(a)
        if self.normalize:
            x = self.layer_norm(x)
        return x.transpose(0, 1), {'attn': attns, 'inner_states': inner}

    def get_normalized_probs(self, net_output, log_probs, sample=None):
        out = self.adaptive_softmax.get_log_prob(net_output[0])               # :160-173
        return out if log_probs else out.exp()

    def filter_incremental_state(self, incremental_state, active_idx):
        if incremental_state is None:                                          # :175-180
            return
        for key in incremental_state:
            if 'DynamicConv1dTBC' in key:
                incremental_state[key] = incremental_state[key][:, active_idx]
