"""tell/models/baseline_glove.py:22-320 (`baseline_glove`, expt/*/1_lstm_glove) on the MI355X path.

The reference turns the raw article strings into GloVe vectors with spaCy INSIDE the model (:207-220).  That lookup is
data plane (no spaCy / vectors offline); this class starts from what it produces - `context_vectors`, a [B, L, 300]
fp32 tensor of the vectors of the tokens that have one, NaN-padded to the longest article of the batch (:216-220) -
passed either directly or as `metadata[i]['context_vectors']`."""
import math

import torch

from .. import ops
from .transformer import CaptionModel, Model


@Model.register('baseline_glove')
class BaselineGloveModel(Model):
    def __init__(self, vocab, decoder, criterion, evaluate_mode=False, namespace='bpe', index='roberta',
                 padding_value=1, use_context=True, sampling_topk=1, sampling_temp=1.0, max_caption_len=50,
                 weigh_bert=False, initializer=None, resnet=None):
        super().__init__(vocab)
        self.decoder, self.criterion = decoder, criterion
        self.index, self.namespace = index, namespace
        if resnet is None:
            from .resnet import resnet152
            resnet = resnet152()
        self.resnet = resnet
        self.use_context, self.padding_idx, self.evaluate_mode = use_context, padding_value, evaluate_mode
        if sampling_topk != 1:
            raise NotImplementedError('generation is greedy (sampling_topk: 1 in every config)')
        self.sampling_topk, self.sampling_temp, self.max_caption_len = sampling_topk, sampling_temp, max_caption_len
        self.n_batches = self.n_samples = 0

    @staticmethod
    def _vectors(context_vectors, metadata):
        if context_vectors is not None:
            return context_vectors
        vs = [torch.as_tensor(m['context_vectors'], dtype=torch.float32) for m in metadata]
        L = max(v.shape[0] for v in vs)
        out = torch.full((len(vs), L, 300), float('nan'))
        for i, v in enumerate(vs):
            out[i, :v.shape[0]] = v
        return out

    def _forward(self, context_vectors, image, caption):                       # :164-245
        dtype = ops.rt.compute_dtype()
        cap = caption[self.index]
        target_ids = torch.zeros_like(cap)
        target_ids[:, :-1] = cap[:, 1:]
        caption_ids = cap[:, :-1][:, :self.max_caption_len].contiguous()       # :176-181
        target_ids = target_ids[:, :-1][:, :self.max_caption_len].contiguous()
        caption[self.index] = caption_ids
        with torch.no_grad():
            x_image = CaptionModel._run_resnet(self, image)                    # [B, 49, 2048]  (:186-198)
        B, P, _ = x_image.shape
        cv = context_vectors.to(image.device)
        Bc, L, dim = cv.shape
        clean = torch.empty(Bc, L, dim, dtype=dtype, device=cv.device)         # :222-226 NaN rows -> mask, zeros
        mask = torch.empty(Bc, L, dtype=torch.uint8, device=cv.device)
        ops.call('tell_nan_rows', cv.float().contiguous(), Bc * L, dim, clean, ops.hip.dt(dtype), mask)
        contexts = {'image': x_image.transpose(0, 1),
                    'image_mask': torch.zeros(B, P, dtype=torch.bool, device=image.device),
                    'article': clean.transpose(0, 1), 'article_mask': mask.bool(),
                    'sections': None, 'sections_mask': None}
        return caption_ids, target_ids, contexts

    def forward(self, image, caption, metadata=None, context_vectors=None):    # :71-162
        ops.hip.require_gpu()
        caption_ids, target_ids, contexts = self._forward(self._vectors(context_vectors, metadata), image, caption)
        ops.rt.wait_weight_update()
        decoder_out = self.decoder(caption, contexts)
        loss_sum, sample_size = self.criterion(self.decoder.adaptive_softmax, decoder_out, target_ids)
        loss = (loss_sum / math.log(2) / sample_size.to(torch.float32)).reshape(())
        out = {'loss': loss, 'sample_size': sample_size.reshape(())}
        if not self.training and self.evaluate_mode:
            _, gen_ids = self._generate(caption_ids, contexts)
            out['gen_ids'] = gen_ids.cpu().numpy()
        self.n_samples += caption_ids.shape[0]
        self.n_batches += 1
        return out

    def generate(self, image, caption, metadata=None, context_vectors=None):
        caption_ids, _, contexts = self._forward(self._vectors(context_vectors, metadata), image, caption)
        log_probs, gen_ids = self._generate(caption_ids, contexts)
        return {'gen_ids': gen_ids, 'log_probs': log_probs}

    @torch.no_grad()
    def _generate(self, caption_ids, contexts, gen_len=100, eos=2):
        """Greedy decode of :247-320.  The reference re-decodes the whole prefix for the still-active rows at every
        step; rows are independent and the decoder is recurrent, so carrying the LSTM state forward and keeping the
        batch at its full size (finished rows masked) gives the same token ids: pad after <eos>, length = 1 + steps
        until the last row has finished."""
        B, dev = caption_ids.shape[0], caption_ids.device
        cur = caption_ids[:, 0:1].contiguous()
        finished = cur[:, 0] == eos
        ids = torch.full((B, gen_len + 1), self.padding_idx, dtype=torch.long, device=dev)
        ids[:, 0] = cur[:, 0]
        lps = torch.zeros(B, gen_len, dtype=torch.float32, device=dev)
        state, steps = {}, gen_len
        for i in range(gen_len):
            out = self.decoder({self.index: cur}, contexts, incremental_state=state)
            lp_all = self.decoder.get_normalized_probs((out[0][:, -1:], None), log_probs=True).squeeze(1).float()
            lp, tok = lp_all.max(dim=-1)
            ids[:, i + 1] = torch.where(finished, ids[:, i + 1], tok)
            lps[:, i] = torch.where(finished, lps[:, i], lp / self.sampling_temp)
            finished = finished | (tok == eos)
            cur = tok.view(B, 1)
            if bool(finished.all()):
                steps = i + 1
                break
        return lps[:, :steps], ids[:, :steps + 1]


@Model.register('transformer_glove')
class TransformerGloveModel(CaptionModel):
    """tell/models/transformer_glove.py (`transformer_glove`, expt/*/2_transformer_glove): the 2-context DynamicConv decoder
    over ResNet regions and GloVe article vectors.  `_forward` is the baseline's (:161-230, no caption truncation);
    generation is the transformer models' (projected-K/V cache, static batch, captured decode step)."""

    def __init__(self, vocab, decoder, criterion, evaluate_mode=False, attention_dim=1024, hidden_size=1024, dropout=0.1,
                 vocab_size=50264, model_name='roberta-base', namespace='bpe', index='roberta', padding_value=1,
                 use_context=True, sampling_topk=1, sampling_temp=1.0, initializer=None, resnet=None):
        Model.__init__(self, vocab)
        self.decoder, self.criterion = decoder, criterion
        self.index, self.namespace = index, namespace
        if resnet is None:
            from .resnet import resnet152
            resnet = resnet152()
        self.resnet = resnet
        self.use_context, self.padding_idx, self.evaluate_mode = use_context, padding_value, evaluate_mode
        if sampling_topk != 1:
            raise NotImplementedError('generation is greedy (sampling_topk: 1 in every config)')
        self.sampling_topk, self.sampling_temp = sampling_topk, sampling_temp
        self.weigh_bert = False
        self.max_caption_len = 1 << 30
        self.n_batches = self.n_samples = 0

    _vectors = staticmethod(BaselineGloveModel._vectors)
    _glove_forward = BaselineGloveModel._forward

    def forward(self, image, caption, metadata=None, context_vectors=None):
        ops.hip.require_gpu()
        caption_ids, target_ids, contexts = self._glove_forward(self._vectors(context_vectors, metadata), image, caption)
        contexts = {k: v for k, v in contexts.items() if v is not None}
        ops.rt.wait_weight_update()
        decoder_out = self.decoder(caption, contexts)
        loss_sum, sample_size = self.criterion(self.decoder.adaptive_softmax, decoder_out, target_ids)
        loss = (loss_sum / math.log(2) / sample_size.to(torch.float32)).reshape(())
        out = {'loss': loss, 'sample_size': sample_size.reshape(())}
        if not self.training and self.evaluate_mode:
            _, gen_ids, _ = self._generate(caption_ids, contexts)
            out['gen_ids'] = gen_ids.cpu().numpy()
        self.n_samples += caption_ids.shape[0]
        self.n_batches += 1
        return out

    def generate(self, image, caption, metadata=None, context_vectors=None, beam_size=1):
        caption_ids, _, contexts = self._glove_forward(self._vectors(context_vectors, metadata), image, caption)
        contexts = {k: v for k, v in contexts.items() if v is not None}
        log_probs, gen_ids, attns = self._generate(caption_ids, contexts, beam_size=beam_size)
        return {'gen_ids': gen_ids, 'log_probs': log_probs, 'attns': attns}
