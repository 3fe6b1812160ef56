"""tell/models/decoder_faces_objects.py:22-379 and tell/models/decoder_flattened.py:23-333
on the MI355X path.  The two reference files are the same block structure over a
different list of contexts; here one table-driven implementation answers to both
registration names and keeps both sets of constructor arguments / state_dict keys."""
import torch
import torch.nn as nn

import os

from .. import blocks, decode, ops
from ..common.registrable import Registrable
from ..modules import AdaptiveSoftmax, DynamicConv1dTBC, GehringLinear, LightweightConv1dTBC, MultiHeadAttention
from ..modules.token_embedders import AdaptiveEmbedding


_BLOCKS = os.environ.get('TELL_BLOCKS', '1') != '0'      # A/B aid: 0 = the per-op composition everywhere


def eval_str_list(x, type=float):
    """tell/utils/options.py:1-9"""
    if x is None:
        return None
    if isinstance(x, str):
        x = eval(x)
    try:
        return list(map(type, x))
    except TypeError:
        return [type(x)]


class Decoder(nn.Module, Registrable):
    pass


class DecoderLayer(nn.Module, Registrable):
    pass


class DynamicConvDecoderLayer(DecoderLayer):
    """decoder_faces_objects.py:184-372 / decoder_flattened.py:185-326 (post-LN only:
    every expt/ config has decoder_normalize_before: false)."""

    def __init__(self, decoder_embed_dim, decoder_conv_dim, decoder_glu, decoder_conv_type, weight_softmax,
                 decoder_attention_heads, weight_dropout, dropout, relu_dropout, input_dropout,
                 decoder_normalize_before, attention_dropout, decoder_ffn_embed_dim, contexts, kernel_size=0):
        super().__init__()
        if decoder_conv_type not in ('dynamic', 'lightweight'):
            raise NotImplementedError('decoder_conv_type must be dynamic or lightweight (decoder_faces_objects.py:199-211)')
        E = self.embed_dim = decoder_embed_dim
        self.conv_dim = decoder_conv_dim
        self.glu = bool(decoder_glu)                                                      # :190-197
        self.linear1 = GehringLinear(E, 2 * self.conv_dim if self.glu else self.conv_dim)
        conv_cls = DynamicConv1dTBC if decoder_conv_type == 'dynamic' else LightweightConv1dTBC      # :199-211
        self.conv = conv_cls(self.conv_dim, kernel_size, padding_l=kernel_size - 1, weight_softmax=weight_softmax,
                             num_heads=decoder_attention_heads, weight_dropout=weight_dropout)
        self.linear2 = GehringLinear(self.conv_dim, E)
        self.dropout, self.relu_dropout, self.input_dropout = dropout, relu_dropout, input_dropout
        self.normalize_before = decoder_normalize_before
        self.conv_layer_norm = nn.LayerNorm(E)
        self.context_attns = nn.ModuleDict()
        self.context_attn_lns = nn.ModuleDict()
        self.context_names = [n for n, _ in contexts]
        for name, kdim in contexts:
            self.context_attns[name] = MultiHeadAttention(E, decoder_attention_heads, kdim=kdim, vdim=kdim,
                                                          dropout=attention_dropout)
            self.context_attn_lns[name] = nn.LayerNorm(E)
        self.context_fc = GehringLinear(E * len(contexts), E)
        self.fc1 = GehringLinear(E, decoder_ffn_embed_dim)
        self.fc2 = GehringLinear(decoder_ffn_embed_dim, E)
        self.final_layer_norm = nn.LayerNorm(E)
        self.need_attn = False      # attention-weight export is opt-in (make_generation_fast_, :374-375)

    @staticmethod
    def _ln(ln, x, res, p, training):
        return ops.layer_norm(x, res, ln.weight, ln.bias, ln.eps, p, training)

    def _pre(self, ln, x):
        """maybe_layer_norm(before=True), :367-372: LayerNorm in front of the block for pre-LN layers."""
        return ops.layer_norm(x, None, ln.weight, ln.bias, ln.eps, 0.0, False) if self.normalize_before else x

    def _post(self, ln, h, res):
        """res + dropout(h), then maybe_layer_norm(after=True): post-LN layers (every expt/ config) take the fused
        LN(res + dropout(h)) kernel."""
        if self.normalize_before:
            return res + ops.dropout(h, self.dropout, self.training)
        return self._ln(ln, h, res, self.dropout, self.training)

    def forward(self, X, contexts, incremental_state, contexts_t=None, kv=None, kv_packed=None):
        tr = self.training
        use_blocks = _BLOCKS and incremental_state is None and blocks.usable(self, X) and X.requires_grad
        if use_blocks:                                                     # :256-266 as one autograd node
            X = blocks.conv_block(self, X)
        else:
            res = X
            h = ops.dropout(self._pre(self.conv_layer_norm, X), self.input_dropout, tr)
            h = self.linear1(h)
            if self.glu:
                h = ops.glu(h)
            h = self.conv(h, incremental_state=incremental_state)
            h = self.linear2(h)
            X = self._post(self.conv_layer_norm, h, res)

        attns, outs = {}, []
        fused = not self.normalize_before and X.is_cuda     # post-LN layers (every expt/ config): LN_c writes its slice
        nctx = len(self.context_names)
        if fused and torch.is_grad_enabled() and X.requires_grad:
            # X feeds the n query projections and (once) the residuals of the n LayerNorms: one fan-in kernel in backward
            handles = ops.fan_out(X, nctx + 1)
        else:
            handles = (X,) * (nctx + 1)
        grouped = (fused and nctx > 1 and ops.rt.compute_dtype() == torch.bfloat16 and X.dtype == torch.bfloat16 and
                   not (not tr and self.need_attn))        # (attention-weight export takes the per-context path)
        if grouped:
            # the n attentions only meet at the LayerNorms: their query projections and their output projections each
            # travel as ONE grouped launch (forward, input gradients; the weight gradients join the trainer's queue).
            # (The four attention cores as one launch measured no gain: each is a full round of B*H workgroups whose
            # time is the kernel's fixed latency, not its work.)
            mods = [self.context_attns[n] for n in self.context_names]
            qs = ops.grouped_linear([handles[i] for i in range(nctx)], [m.q_spec() for m in mods])
            cores = None
            if kv is not None and not tr and X.shape[0] == 1 and torch.is_tensor(qs[0]) and qs[0].is_contiguous():
                # one generated token against the projected K / V cache, layer-by-layer step (more rows than the fused
                # step of decode.py takes): the n attention cores as ONE launch of the decode kernel instead of n launches
                # of the training kernel with one real query row each (+ a strided copy per context under beam search)
                from .. import decode as _dec
                q2 = [q.reshape(q.shape[1], q.shape[2]) for q in qs]
                if _dec.attn_decode_usable(mods, kv, self.context_names, q2[0]):
                    a_all = _dec.attn_decode_all(mods, self.context_names, q2, kv, contexts, q2[0].shape[0], q2[0].shape[1])
                    cores = [a_all[i].unsqueeze(0) for i in range(nctx)]
            if cores is None:
                cores = [m.core(q, contexts[n], contexts[n + '_mask'], None if kv is None else kv[n],
                                None if kv_packed is None else kv_packed.get(n))
                         for m, q, n in zip(mods, qs, self.context_names)]
            outs = ops.grouped_linear(cores, [m.out_spec() for m in mods])
        for i, name in enumerate(() if grouped else self.context_names):  # :271-352
            a, w = self.context_attns[name](
                self._pre(self.context_attn_lns[name], handles[i]), contexts[name], contexts[name],
                key_padding_mask=contexts[name + '_mask'], need_weights=(not tr and self.need_attn),
                key_t=None if contexts_t is None else contexts_t.get(name),
                kv=None if kv is None else kv[name])
            outs.append(a if fused else self._post(self.context_attn_lns[name], a, X))
            if w is not None:
                attns[name] = w.cpu().numpy()
        if fused:
            cat = ops.layer_norm_cat(outs, handles[nctx], [self.context_attn_lns[n] for n in self.context_names],
                                     self.dropout, tr)
        else:
            cat = torch.cat(outs, dim=-1)
        X = self.context_fc(cat)                                          # :354-355
        if use_blocks:
            return blocks.ffn_block(self, X), attns                        # :357-364 as one autograd node

        res = X                                                            # :357-364
        h = self.fc1(self._pre(self.final_layer_norm, X), act=1)
        h = ops.dropout(h, self.relu_dropout, tr)
        h = self.fc2(h)
        return self._post(self.final_layer_norm, h, res), attns

    def make_generation_fast_(self, need_attn=False, **kwargs):
        self.need_attn = need_attn


class _DynamicConvDecoderBase(Decoder):
    CONTEXTS = ()

    def __init__(self, vocab, embedder, max_target_positions, dropout, share_decoder_input_output_embed,
                 decoder_output_dim, decoder_conv_dim, decoder_glu, decoder_conv_type, weight_softmax,
                 decoder_attention_heads, weight_dropout, relu_dropout, input_dropout, decoder_normalize_before,
                 attention_dropout, decoder_ffn_embed_dim, decoder_kernel_size_list, adaptive_softmax_cutoff=None,
                 tie_adaptive_weights=False, adaptive_softmax_dropout=0, tie_adaptive_proj=False,
                 adaptive_softmax_factor=0, decoder_layers=6, final_norm=True, padding_idx=0,
                 namespace='target_tokens', vocab_size=None, section_attn=False, swap=False,
                 article_embed_size=1024):
        super().__init__()
        self.vocab = vocab
        vocab_size = vocab_size or vocab.get_vocab_size(namespace)
        self.dropout = dropout
        self.share_input_output_embed = share_decoder_input_output_embed
        E = embedder.get_output_dim()
        self.max_target_positions = max_target_positions
        self.embedder = embedder
        self.project_in_dim = None
        contexts = tuple((n, article_embed_size if n == 'article' and self.ARTICLE_DIM_FROM_ARG else k)
                         for n, k in self.CONTEXTS)
        self.layers = nn.ModuleList([
            DynamicConvDecoderLayer(E, decoder_conv_dim, decoder_glu, decoder_conv_type, weight_softmax,
                                    decoder_attention_heads, weight_dropout, dropout, relu_dropout,
                                    input_dropout, decoder_normalize_before, attention_dropout,
                                    decoder_ffn_embed_dim, contexts, kernel_size=decoder_kernel_size_list[i])
            for i in range(decoder_layers)])
        if adaptive_softmax_cutoff is None or not tie_adaptive_weights:
            raise NotImplementedError('HIP path implements the tied adaptive-softmax head of the configs')
        adaptive_inputs = embedder if isinstance(embedder, AdaptiveEmbedding) else embedder.token_embedder_adaptive
        self.project_out_dim = None
        self.adaptive_softmax = AdaptiveSoftmax(vocab_size, E, eval_str_list(adaptive_softmax_cutoff, type=int),
                                                dropout=adaptive_softmax_dropout, adaptive_inputs=adaptive_inputs,
                                                factor=adaptive_softmax_factor, tie_proj=tie_adaptive_proj)
        self.register_buffer('version', torch.Tensor([2]))
        self.normalize = decoder_normalize_before and final_norm
        if self.normalize:                                                         # :88-90
            self.layer_norm = nn.LayerNorm(E)

    def _wn_pairs(self):
        pairs = getattr(self, '_wn_pair_list', None)
        if pairs is None:
            pairs = [(m.weight_g, m.weight_v) for m in self.modules() if isinstance(m, GehringLinear)]
            object.__setattr__(self, '_wn_pair_list', pairs)
        return pairs

    def forward(self, prev_target, contexts, incremental_state=None, use_layers=None, kv_cache=None, **kwargs):
        if self.training and torch.is_grad_enabled():
            ops.wn_prepare(self._wn_pairs())          # every weight-normalised working weight of the step in one launch
        X = self.embedder(prev_target, incremental_state=incremental_state)      # :98  [B,T,E] view
        if incremental_state is not None and incremental_state.get('_ring'):
            # index of this step for the DynamicConv rings (host part; a captured step adds the device counter)
            incremental_state['_t_cur'] = incremental_state.get('_t', 0)
            incremental_state['_t'] = incremental_state['_t_cur'] + 1
        X = X.transpose(0, 1)                                                      # :109 T x B x C (contiguous)
        X = ops.dropout(X, self.dropout, self.training)                            # :106
        contexts_t = None
        if torch.is_grad_enabled() and self.training and ops.rt.compute_dtype() == torch.float32:
            # fp32 parity mode only (the bf16 wgrad GEMM reads K-major operands in place): one transpose per
            # context per step, shared by the K and V weight-gradient GEMMs of all layers
            contexts_t = contexts.get('_transposed')
            if contexts_t is None:
                contexts_t = {}
                for name, _ in self.CONTEXTS:
                    c = contexts[name]
                    if c.shape[0] > 0 and c.shape[2] > 0:
                        bm = c.transpose(0, 1)
                        src = bm if (bm.is_contiguous() and not c.is_contiguous()) else c.contiguous()
                        contexts_t[name] = ops.transpose(ops.as2d(src))[0]
                contexts['_transposed'] = contexts_t
        kv_packed = None
        if (_BLOCKS and self.training and torch.is_grad_enabled() and X.is_cuda and X.requires_grad and
                incremental_state is None and kv_cache is None and ops.rt.compute_dtype() == torch.bfloat16 and
                len(self.CONTEXTS) > 1 and not any(l.normalize_before or l.need_attn for l in self.layers) and
                ops.rt.grad_ready_callback() is None):      # (a bucketed DP exchange needs a layer's gradients final
                                                            #  when backward leaves the layer: no cross-layer node then)
            # nothing below depends on the decoder state: the W^T copies the backward pass will want, and the K|V
            # projections of every (layer, context) pair, as one launch each before the layer loop
            blocks.prepare_transposes(self.layers, X.shape[0] * X.shape[1])
            kv_packed = blocks.kv_project_all(self.layers, [n for n, _ in self.CONTEXTS], contexts)
        attns, inner_states = [], [X]
        if X.is_cuda:                        # key-padding masks as the uint8 the attention kernels read: once, not per layer
            contexts = dict(contexts)
            for name, _ in self.CONTEXTS:
                m = contexts.get(name + '_mask')
                if torch.is_tensor(m) and m.dtype == torch.bool:
                    contexts[name + '_mask'] = m.to(torch.uint8).contiguous()
        if not use_layers and decode.usable(self, X, incremental_state, kv_cache):
            # generation: the whole layer stack of this step as weight-streaming launches (decode.py)
            X = decode.decoder_step(self, X, contexts, incremental_state, kv_cache)
            return X.transpose(0, 1), {'attn': [], 'inner_states': []}
        for i, layer in enumerate(self.layers):
            if not use_layers or i in use_layers:
                X = ops.grad_ready_marker(X, 'decoder.layers.%d.' % i)    # DP: layer i's gradients are final here
                X, attn = layer(X, contexts, incremental_state, contexts_t,
                                None if kv_cache is None else kv_cache[i], None if kv_packed is None else kv_packed[i])
                inner_states.append(X)
            attns.append(attn)
        if self.normalize:                                                         # :125-126
            X = ops.layer_norm(X, None, self.layer_norm.weight, self.layer_norm.bias, self.layer_norm.eps, 0.0, False)
        X = X.transpose(0, 1)                                                      # :129 B x T x C
        return X, {'attn': attns, 'inner_states': inner_states}

    def project_contexts(self, contexts):
        """Per layer, per context: the K and V projections (multi_head.py:500-518).  The reference
        recomputes these 12.4 GFLOP/sample for every generated token (decoder_faces_objects.py:280
        passes incremental_state=None); they only depend on the static contexts."""
        return [{name: layer.context_attns[name].project_kv(contexts[name]) for name in layer.context_names}
                for layer in self.layers]

    def max_positions(self):
        return self.max_target_positions

    def get_normalized_probs(self, net_output, log_probs, sample=None):            # :160-173
        out = self.adaptive_softmax.get_log_prob(net_output[0])
        return out if log_probs else out.exp()

    def static_incremental_state(self, batch, device, dtype, beam=False):
        """Incremental state of fixed shape for the captured decode step.  Where every layer's convolution takes the step
        kernel (bf16, 64-wide heads: csrc/decode.hip tell_dynconv_step) the input buffers are RINGS: K planes indexed by
        time, never shifted, and - beam search - never re-ordered: `_back` [Kmax-1, batch] int32 says in which slot a
        hypothesis' past rows lie (reorder_incremental_state composes it with the parents).  Otherwise (fp32 parity mode,
        other convolution types) every buffer is K-1 zero rows in time order, shifted and re-ordered physically."""
        st = {'_static': True}
        convs = [layer.conv for layer in self.layers]
        ring = (dtype == torch.bfloat16 and torch.device(device).type == 'cuda' and
                all(hasattr(c, 'ring_usable') and c.ring_usable() for c in convs))
        for conv in convs:
            st[conv._state_key] = torch.zeros(conv.kernel_size - (0 if ring else 1), batch, conv.input_size, dtype=dtype,
                                              device=device)
        if ring:
            st['_ring'] = True
            if beam:
                kmax = max(c.kernel_size for c in convs)
                st['_back'] = torch.arange(batch, dtype=torch.int32, device=device).repeat(max(kmax - 1, 1), 1).contiguous()
        return st

    @staticmethod
    def reset_static_state(st):
        """A new caption batch on the same buffers: empty history, step 0."""
        for k, s in st.items():
            if torch.is_tensor(s):
                if k == '_back':
                    s.copy_(torch.arange(s.shape[1], dtype=torch.int32, device=s.device).expand_as(s))
                else:
                    s.zero_()
        st.pop('_t', None)
        st.pop('_t_cur', None)

    def reorder_incremental_state(self, incremental_state, new_order):
        """Beam search: row r of the new state is row new_order[r] of the old one (dynamic.py:338-342)."""
        if incremental_state is None:
            return
        if incremental_state.get('_ring'):
            # rings: nothing moves - slot r's past is where its parent's past was; one step ago it sat in the parent's slot
            back = incremental_state.get('_back')
            if back is None:
                raise RuntimeError('ring-buffer decode state without an ancestor table: create it with beam=True')
            order = new_order.to(torch.int32)
            nb = torch.empty_like(back)
            nb[0] = order
            if back.shape[0] > 1:
                nb[1:] = back[:-1].index_select(1, new_order)
            back.copy_(nb)
            return
        for key in incremental_state:
            if 'Conv1dTBC' in key:
                incremental_state[key] = incremental_state[key].index_select(1, new_order)

    def filter_incremental_state(self, incremental_state, active_idx):             # :175-180
        if incremental_state is None:
            return
        for key in incremental_state:
            if 'Conv1dTBC' in key:          # (the reference names DynamicConv1dTBC only)
                incremental_state[key] = incremental_state[key][:, active_idx]


@Decoder.register('dynamic_conv_decoder_faces_objects')
class DynamicConvFacesObjectsDecoder(_DynamicConvDecoderBase):
    """tell/models/decoder_faces_objects.py:22 (`contexts['image'|'article'|'faces'|'obj']`)."""
    CONTEXTS = (('image', 2048), ('article', 1024), ('faces', 512), ('obj', 2048))
    ARTICLE_DIM_FROM_ARG = False


@Decoder.register('dynamic_conv_decoder_flattened')
class DynamicConvDecoder(_DynamicConvDecoderBase):
    """tell/models/decoder_flattened.py:23 (`contexts['image'|'article']`)."""
    CONTEXTS = (('image', 2048), ('article', 1024))
    ARTICLE_DIM_FROM_ARG = True


@Decoder.register('dynamic_conv_decoder_faces_parallel')
class DynamicConvFacesParallelDecoder(_DynamicConvDecoderBase):
    """tell/models/decoder_faces_parallel.py:22 (`contexts['image'|'article'|'faces']`; expt/*/8_transformer_faces
    and the copying ablations): the faces+objects layer without the object attention (context_size 3E, :240)."""
    CONTEXTS = (('image', 2048), ('article', 1024), ('faces', 512))
    ARTICLE_DIM_FROM_ARG = False


@Decoder.register('dynamic_conv_decoder_flattened_no_image')
class DynamicConvDecoderNoImage(_DynamicConvDecoderBase):
    """tell/models/decoder_flattened_no_image.py:22 (`contexts['article']` only; expt/*/4_no_image)."""
    CONTEXTS = (('article', 1024),)
    ARTICLE_DIM_FROM_ARG = True
