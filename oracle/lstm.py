"""LSTM decoder of the GloVe/LSTM baseline for the oracle.

TEST INFRASTRUCTURE - see oracle/__init__.py.

Restates tell/models/decoder_flattened_lstm.py: `AttentionLayer` :29-65 (weight-normalised input projection, dot-product
scores against the source states, key-padding mask, softmax over the source length, weighted sum, tanh(output projection of
[context ; input])) and `LSTMDecoder` :68-208 (learned initial states, `num_layers` LSTM cells with input feeding - the
attention output of step t-1 is concatenated to the token embedding of step t -, one attention over the image regions and one
over the article per step, `attn_proj`, optional `project_out_dim`, tied adaptive softmax)."""
import torch
import torch.nn as nn

from .modules import AdaptiveSoftmax, GehringLinear, _maybe_dropout


class AttentionLayer(nn.Module):
    def __init__(self, input_embed_dim, source_embed_dim, output_embed_dim, bias=False):
        super().__init__()
        self.input_proj = GehringLinear(input_embed_dim, source_embed_dim, bias=bias)
        self.output_proj = GehringLinear(input_embed_dim + source_embed_dim, output_embed_dim, bias=bias)

    def forward(self, inp, source_hids, padding_mask):
        x = self.input_proj(inp)                                           # [B, D_src]
        scores = (source_hids * x.unsqueeze(0)).sum(dim=2)                 # [L, B]           (:46)
        scores = scores.masked_fill(padding_mask.transpose(0, 1), float('-inf'))             # (:53-57)
        probs = torch.softmax(scores, dim=0)                               # over the source length (:59)
        ctx = (probs.unsqueeze(2) * source_hids).sum(dim=0)                # [B, D_src]       (:62)
        return torch.tanh(self.output_proj(torch.cat((ctx, inp), dim=1))), probs


def _lstm_cell(input_size, hidden_size):
    m = nn.LSTMCell(input_size, hidden_size)
    for name, p in m.named_parameters():                                   # :20-26
        if 'weight' in name or 'bias' in name:
            p.data.uniform_(-0.1, 0.1)
    return m


class LSTMDecoder(nn.Module):
    def __init__(self, embedder, num_layers, hidden_size, dropout, vocab_size, adaptive_softmax_cutoff,
                 article_embed_size=1024, image_embed_size=2048):
        super().__init__()
        self.embedder, self.hidden_size, self.dropout = embedder, hidden_size, dropout
        E = embedder.get_output_dim()
        self.layers = nn.ModuleList([_lstm_cell(hidden_size + E if i == 0 else hidden_size, hidden_size)
                                     for i in range(num_layers)])
        self.h = nn.ParameterList([nn.Parameter(torch.zeros(1, hidden_size)) for _ in range(num_layers)])
        self.c = nn.ParameterList([nn.Parameter(torch.zeros(1, hidden_size)) for _ in range(num_layers)])
        self.image_attention = AttentionLayer(hidden_size, image_embed_size, hidden_size, bias=True)
        self.article_attention = AttentionLayer(hidden_size, article_embed_size, hidden_size, bias=True)
        self.attn_proj = GehringLinear(hidden_size * 2, hidden_size)
        self.project_out_dim = GehringLinear(hidden_size, E, bias=False) if hidden_size != E else None
        self.adaptive_softmax = AdaptiveSoftmax(vocab_size, E, list(adaptive_softmax_cutoff),
                                                embedder.token_embedder_adaptive)

    def forward(self, prev_target, contexts, incremental_state=None):
        tr = self.training
        x = self.embedder(prev_target, incremental_state=incremental_state)
        x = _maybe_dropout(x, self.dropout, tr).transpose(0, 1)            # T x B x C (:137-141)
        T, B, _ = x.shape
        hs = [h.expand(B, -1) for h in self.h]
        cs = [c.expand(B, -1) for c in self.c]
        feed = x.new_zeros(B, self.hidden_size)
        outs = []
        for t in range(T):                                                 # :155-186
            inp = torch.cat((x[t], feed), dim=1)
            for i, cell in enumerate(self.layers):
                hs[i], cs[i] = cell(inp, (hs[i], cs[i]))
                inp = _maybe_dropout(hs[i], self.dropout, tr)
            img, _ = self.image_attention(hs[-1], contexts['image'], contexts['image_mask'])
            art, _ = self.article_attention(hs[-1], contexts['article'], contexts['article_mask'])
            feed = self.attn_proj(_maybe_dropout(torch.cat([img, art], dim=1), self.dropout, tr))
            outs.append(feed)
        y = torch.stack(outs, dim=0).transpose(0, 1)                       # B x T x H
        if self.project_out_dim is not None:
            y = self.project_out_dim(y)
        return y, {'attn': None, 'inner_states': None}


class BaselineGloveModel(nn.Module):
    """tell/models/baseline_glove.py:22-320 from the point where the article is a NaN-padded tensor of GloVe vectors
    (`context_vectors` [B,L,300], what :207-220 builds with spaCy): caption shift + truncation (:168-183), ResNet regions
    (:186-198), NaN rows -> padding mask and zeros (:222-226), LSTM decoder, adaptive loss in bits per token (:80-90),
    greedy generation by re-decoding the prefix of the still-active rows (:247-320)."""

    def __init__(self, decoder, criterion, resnet, padding_value=1, max_caption_len=50, sampling_temp=1.0):
        super().__init__()
        self.decoder, self.criterion, self.resnet = decoder, criterion, resnet
        self.padding_idx, self.max_caption_len, self.sampling_temp = padding_value, max_caption_len, sampling_temp

    def _forward(self, context_vectors, image, caption_ids_full):
        import math  # noqa: F401
        cap = caption_ids_full
        target = torch.zeros_like(cap)
        target[:, :-1] = cap[:, 1:]
        cap, target = cap[:, :-1][:, :self.max_caption_len], target[:, :-1][:, :self.max_caption_len]
        x = self.resnet(image).permute(0, 2, 3, 1)
        B, H, W, C = x.shape
        x = x.reshape(B, H * W, C)
        cv = context_vectors.clone()
        mask = torch.isnan(cv).any(dim=-1)
        cv[mask] = 0
        ctx = {'image': x.transpose(0, 1), 'image_mask': torch.zeros(B, H * W, dtype=torch.bool),
               'article': cv.transpose(0, 1), 'article_mask': mask}
        return cap, target, ctx

    def forward(self, image, caption_ids, context_vectors):
        import math
        cap, target, ctx = self._forward(context_vectors, image, caption_ids)
        out = self.decoder({'roberta': cap}, ctx)
        loss, n = self.criterion(self.decoder.adaptive_softmax, out, target)
        return {'loss': loss / math.log(2) / n, 'sample_size': n}

    @torch.no_grad()
    def generate(self, image, caption_ids, context_vectors, gen_len=100, eos=2):
        cap, _, ctx = self._forward(context_vectors, image, caption_ids)
        B = cap.shape[0]
        seed = cap[:, 0:1]
        active = seed[:, -1] != eos                       # rows still decoding (indices into the full batch)
        paths, lps = [seed], []
        for _ in range(gen_len):
            sub = {k: (v[:, active] if k in ('image', 'article') else v[active]) for k, v in ctx.items()}
            out = self.decoder({'roberta': seed}, sub)
            lp = self.decoder.adaptive_softmax.get_log_prob(out[0][:, -1:]).squeeze(1)
            top_lp, top = lp.max(dim=-1)
            path = torch.full((B, 1), self.padding_idx, dtype=torch.long)
            path[active] = top.unsqueeze(1)
            lpf = torch.zeros(B, 1)
            lpf[active] = (top_lp / self.sampling_temp).unsqueeze(1)
            paths.append(path)
            lps.append(lpf)
            seed = torch.cat([seed, top.unsqueeze(1)], dim=-1)
            still = top != eos
            idx = active.nonzero().squeeze(1)
            active = active.clone()
            active[idx[~still]] = False
            seed = seed[still]
            if not bool(active.any()):
                break
        return torch.cat(lps, dim=-1), torch.cat(paths, dim=-1)


class TransformerGloveModel(BaselineGloveModel):
    """tell/models/transformer_glove.py (`transformer_glove`): the baseline's `_forward` without the caption truncation
    (:161-230) in front of the 2-context DynamicConv decoder; generation is transformer_flattened's incremental greedy
    decode (`oracle.models.CaptionModel._generate`)."""

    def __init__(self, decoder, criterion, resnet, padding_value=1, sampling_temp=1.0):
        super().__init__(decoder, criterion, resnet, padding_value, max_caption_len=1 << 30, sampling_temp=sampling_temp)
        self.index, self.sampling_topk = 'roberta', 1

    @torch.no_grad()
    def generate(self, image, caption_ids, context_vectors):
        from .models import CaptionModel
        cap, _, ctx = self._forward(context_vectors, image, caption_ids)
        lp, ids, _ = CaptionModel._generate(self, cap, ctx)
        return lp, ids
