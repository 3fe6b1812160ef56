"""Model wrappers (training forward + loss, greedy generation) for the oracle.

TEST INFRASTRUCTURE - see oracle/__init__.py.

Restates tell/models/transformer_faces_objects.py:67-140,311-494 and
tell/models/transformer_flattened.py:72-142,166-330.  The two encoders are
injected (the reference builds them from torchvision / torch.hub, both absent
here); `oracle.encoders` holds architecture restatements of them.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


class CaptionModel(nn.Module):
    """`use_faces_objects=False` == TransformerFlattenedModel,
    `use_faces_objects=True`  == TransformerFacesObjectModel."""

    def __init__(self, decoder, criterion, resnet, roberta, use_faces_objects, weigh_bert=True,
                 n_bert_layers=25, padding_value=1, index='roberta', evaluate_mode=False,
                 sampling_topk=1, sampling_temp=1.0):
        super().__init__()
        self.decoder, self.criterion = decoder, criterion
        self.resnet, self.roberta = resnet, roberta
        self.use_faces_objects = use_faces_objects
        self.padding_idx, self.index = padding_value, index
        self.evaluate_mode = evaluate_mode
        self.sampling_topk, self.sampling_temp = sampling_topk, sampling_temp
        self.weigh_bert = weigh_bert
        if weigh_bert:
            self.bert_weight = nn.Parameter(torch.rand(n_bert_layers))   # nn.init.uniform_, :57-59
        self.n_batches = self.n_samples = 0

    # ---- transformer_faces_objects.py:311-397 ------------------------------
    def _forward(self, context, image, caption, face_embeds=None, obj_embeds=None):
        cap = caption[self.index]
        target_ids = cap[:, 1:].clone()                                   # :321-328
        caption_ids = cap[:, :-1]
        caption[self.index] = caption_ids                                 # :329 (in-place on the dict)

        feat = self.resnet(image)                                         # :332 [B,2048,7,7]
        B, C, Hh, Ww = feat.shape
        x_image = feat.permute(0, 2, 3, 1).reshape(B, Hh * Ww, C)         # :335-341

        article_ids = context[self.index]
        article_mask = article_ids == self.padding_idx                    # :347
        hiddens = self.roberta.extract_features(article_ids, return_all_hiddens=True)  # :352
        if self.weigh_bert:                                               # :355-364
            w = F.softmax(self.bert_weight, dim=0)
            x_article = (torch.stack(hiddens, dim=2) * w[None, None, :, None]).sum(dim=2)
        else:
            x_article = hiddens[-1]

        contexts = {
            'image': x_image.transpose(0, 1),
            'image_mask': torch.zeros(B, Hh * Ww, dtype=torch.bool),      # :371
            'article': x_article.transpose(0, 1),
            'article_mask': article_mask,
        }
        if self.use_faces_objects:                                        # :373-379, :390-393
            fm = torch.isnan(face_embeds).any(dim=-1)
            face_embeds[fm] = 0
            contexts.update(faces=face_embeds.transpose(0, 1), faces_mask=fm)
            if obj_embeds is not None:                                    # transformer_faces.py has no objects
                om = torch.isnan(obj_embeds).any(dim=-1)
                obj_embeds[om] = 0
                contexts.update(obj=obj_embeds.transpose(0, 1), obj_mask=om)
        return caption_ids, target_ids, contexts

    # ---- transformer_faces_objects.py:67-140 -------------------------------
    def forward(self, context, image, caption, face_embeds=None, obj_embeds=None, metadata=None,
                names=None, attn_idx=None):
        caption_ids, target_ids, contexts = self._forward(context, image, caption, face_embeds,
                                                          obj_embeds)
        decoder_out = self.decoder(caption, contexts)
        loss, sample_size = self.criterion(self.decoder.adaptive_softmax, decoder_out, target_ids)
        loss = loss / math.log(2)                                          # :85 bits
        out = {'loss': loss / sample_size, 'sample_size': sample_size}
        if not self.training and self.evaluate_mode:
            _, gen_ids, attns = self._generate(caption_ids, contexts)
            out['gen_ids'] = gen_ids.numpy()
            out['attns'] = attns
        self.n_samples += caption_ids.shape[0]
        self.n_batches += 1
        return out

    def generate(self, context, image, caption, face_embeds=None, obj_embeds=None, metadata=None,
                 names=None):
        caption_ids, _, contexts = self._forward(context, image, caption, face_embeds, obj_embeds)
        log_probs, gen_ids, attns = self._generate(caption_ids, contexts)
        return {'gen_ids': gen_ids, 'log_probs': log_probs, 'attns': attns}

    # ---- transformer_faces_objects.py:399-494 ------------------------------
    def _generate(self, caption_ids, contexts, gen_len=100, eos=2):
        """Greedy decode with the reference's active-row compaction: finished
        rows leave the batch; token_ids rows are padded with `padding_idx`."""
        state = {}
        B = caption_ids.shape[0]
        seed = caption_ids[:, 0:1]
        alive = seed[:, -1] != eos                     # rows still decoding, full-batch index
        keep = alive                                   # which rows of the *previous* step survive
        cur = seed
        log_probs, paths, attns = [], [seed], []
        ctx_names = [k for k in contexts if not k.endswith('_mask')]
        for _ in range(gen_len):
            self.decoder.filter_incremental_state(state, keep)             # :417
            ctx_i = {}
            for n in ctx_names:                                            # :420-431
                ctx_i[n] = contexts[n][:, alive]
                ctx_i[n + '_mask'] = contexts[n + '_mask'][alive]
            dec_out = self.decoder({self.index: cur[:, -1:]}, ctx_i, incremental_state=state)
            attns.append(dec_out[1]['attn'])
            lp = self.decoder.get_normalized_probs((dec_out[0][:, -1:], None), log_probs=True)
            lp = lp.squeeze(1)
            top_lp, top_ix = lp.topk(self.sampling_topk)                   # :450
            top_lp = top_lp / self.sampling_temp
            if self.sampling_topk == 1:                                    # multinomial over 1 item
                pick = torch.zeros(top_lp.shape[0], 1, dtype=torch.long)
            else:
                pick = torch.multinomial(top_lp.exp(), 1)
            sel_lp, sel_ix = top_lp.gather(1, pick), top_ix.gather(1, pick)
            full_lp = sel_lp.new_zeros(B, 1)
            full_lp[alive] = sel_lp
            full_ix = sel_ix.new_full((B, 1), self.padding_idx)
            full_ix[alive] = sel_ix
            log_probs.append(full_lp)
            paths.append(full_ix)
            keep = sel_ix.squeeze(-1) != eos                               # :476-483
            alive = alive.clone()
            alive[alive.nonzero().squeeze(1)[~keep]] = False
            cur = torch.cat([cur, sel_ix], dim=1)[keep]
            if int(keep.sum()) == 0:                                       # :485
                break
        return torch.cat(log_probs, dim=-1), torch.cat(paths, dim=-1), attns
