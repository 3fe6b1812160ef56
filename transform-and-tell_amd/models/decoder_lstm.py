"""tell/models/decoder_flattened_lstm.py:68-208 (`lstm_decoder_flattened`, the decoder of the GloVe/LSTM baseline
expt/*/1_lstm_glove) on the MI355X path: same constructor arguments and state_dict keys."""
import torch
import torch.nn as nn

from .. import ops
from ..modules import AdaptiveSoftmax, GehringLinear
from ..modules.lstm import AttentionLayer, LSTMCell
from ..modules.token_embedders import AdaptiveEmbedding
from .decoders import Decoder, eval_str_list


@Decoder.register('lstm_decoder_flattened')
class LSTMDecoder(Decoder):
    def __init__(self, vocab, embedder, num_layers, hidden_size, dropout, share_decoder_input_output_embed,
                 vocab_size=None, adaptive_softmax_cutoff=None, tie_adaptive_weights=False, adaptive_softmax_dropout=0,
                 tie_adaptive_proj=False, adaptive_softmax_factor=0, article_embed_size=1024, image_embed_size=2048,
                 namespace='target_tokens'):
        super().__init__()
        self.vocab = vocab
        self.hidden_size = hidden_size
        vocab_size = vocab_size or vocab.get_vocab_size(namespace)
        self.dropout = dropout
        self.share_input_output_embed = share_decoder_input_output_embed
        E = embedder.get_output_dim()
        self.layers = nn.ModuleList([LSTMCell(hidden_size + E if i == 0 else hidden_size, hidden_size)
                                     for i in range(num_layers)])                                  # :92-100
        self.h = nn.ParameterList([nn.Parameter(torch.zeros(1, hidden_size)) for _ in range(num_layers)])
        self.c = nn.ParameterList([nn.Parameter(torch.zeros(1, hidden_size)) for _ in range(num_layers)])
        self.image_attention = AttentionLayer(hidden_size, image_embed_size, hidden_size, bias=True)
        self.article_attention = AttentionLayer(hidden_size, article_embed_size, hidden_size, bias=True)
        self.attn_proj = GehringLinear(hidden_size * 2, hidden_size)
        self.embedder = embedder
        self.project_out_dim = GehringLinear(hidden_size, E, bias=False) if hidden_size != E else None
        if adaptive_softmax_cutoff is None or not tie_adaptive_weights:
            raise NotImplementedError('HIP path implements the tied adaptive-softmax head of the configs')
        adaptive_inputs = embedder if isinstance(embedder, AdaptiveEmbedding) else embedder.token_embedder_adaptive
        self.adaptive_softmax = AdaptiveSoftmax(vocab_size, E, eval_str_list(adaptive_softmax_cutoff, type=int),
                                                dropout=adaptive_softmax_dropout, adaptive_inputs=adaptive_inputs,
                                                factor=adaptive_softmax_factor, tie_proj=tie_adaptive_proj)

    def forward(self, prev_target, contexts, incremental_state=None, use_layers=None, **kwargs):
        """incremental_state (generation): besides the embedder's position it carries the recurrent state
        (`LSTMDecoder.state`), so a call with the NEXT token continues where the previous call stopped - the same
        arithmetic as re-decoding the whole prefix (baseline_glove.py:252-275 does that), without the O(T^2)."""
        tr = self.training
        X = self.embedder(prev_target, incremental_state=incremental_state)
        X = ops.dropout(X, self.dropout, tr).transpose(0, 1)                  # T x B x C (:137-141)
        T, B, _ = X.shape
        carried = incremental_state.get('LSTMDecoder.state') if incremental_state is not None else None
        if carried is not None:
            hs, cs, feed = list(carried[0]), list(carried[1]), carried[2]
        else:
            hs = [h.to(X.dtype).expand(B, -1).contiguous() for h in self.h]  # :148-149 learned initial states
            cs = [c.to(X.dtype).expand(B, -1).contiguous() for c in self.c]
            feed = X.new_zeros(B, self.hidden_size)
        outs = []
        for t in range(T):                                                    # :155-186
            inp = torch.cat((X[t], feed), dim=1)                              # input feeding
            for i, cell in enumerate(self.layers):
                hs[i], cs[i] = cell(inp, (hs[i], cs[i]))
                inp = ops.dropout(hs[i], self.dropout, tr)
            img, _ = self.image_attention(hs[-1], contexts['image'], contexts['image_mask'])
            art, _ = self.article_attention(hs[-1], contexts['article'], contexts['article_mask'])
            feed = self.attn_proj(ops.dropout(torch.cat([img, art], dim=1), self.dropout, tr))
            outs.append(feed)
        if incremental_state is not None:
            incremental_state['LSTMDecoder.state'] = (hs, cs, feed)
        Y = torch.stack(outs, dim=0).transpose(0, 1)                          # B x T x hidden
        if self.project_out_dim is not None:
            Y = self.project_out_dim(Y)
        return Y, {'attn': None, 'inner_states': None}

    def get_normalized_probs(self, net_output, log_probs, sample=None):
        out = self.adaptive_softmax.get_log_prob(net_output[0])
        return out if log_probs else out.exp()
