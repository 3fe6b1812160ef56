"""Beam search over the oracle decoder.  TEST INFRASTRUCTURE - see oracle/__init__.py.

The reference has no beam search (it samples top-1, transformer_faces_objects.py:443-464); SURVEY.md 8-f1 /
BASELINE.json configs[4] ask for beam-4 with the contract "beam = 1 reproduces the greedy ids".  This is the
plain definition the product is checked against: NO incremental state - the whole prefix is re-decoded at every
step - so it also checks the product's cached K/V + reordered DynamicConv buffers.  Score = sum of token
log-probs, no length penalty; a finished hypothesis keeps its score and is extended with pad."""
import torch


@torch.no_grad()
def beam_search(model, caption_ids, contexts, beam_size, gen_len=100, eos=2):
    """model: oracle CaptionModel; contexts as returned by model._forward.  -> (ids [B, L], scores [B])."""
    B, K, pad = caption_ids.shape[0], beam_size, model.padding_idx
    ctx = {}
    for name, val in contexts.items():
        ctx[name] = val.repeat_interleave(K, dim=0 if name.endswith('_mask') else 1)
    seqs = caption_ids[:, 0:1].repeat_interleave(K, dim=0).view(B, K, 1)
    cum = torch.full((B, K), float('-inf'))
    cum[:, 0] = 0.0
    finished = seqs[:, :, 0] == eos
    for _ in range(gen_len):
        out = model.decoder({model.index: seqs.view(B * K, -1)}, ctx, incremental_state=None)
        lp = model.decoder.get_normalized_probs((out[0][:, -1:], None), log_probs=True).view(B, K, -1)
        lp = lp / model.sampling_temp
        V = lp.shape[-1]
        lp = lp.masked_fill(finished.unsqueeze(-1), float('-inf'))
        lp[..., pad] = torch.where(finished, torch.zeros_like(cum), lp[..., pad])
        top, idx = (cum.unsqueeze(-1) + lp).view(B, K * V).topk(K, dim=1)
        parent, tok = idx // V, idx % V
        was = finished.gather(1, parent)
        tok = torch.where(was, torch.full_like(tok, pad), tok)
        seqs = torch.cat([seqs.gather(1, parent.unsqueeze(-1).expand(-1, -1, seqs.shape[2])), tok.unsqueeze(-1)], 2)
        finished = was | (tok == eos)
        cum = top
        if bool(finished.all()):
            break
    return seqs[:, 0], cum[:, 0]
