"""tell/modules/softmax.py:43-222 on the MI355X path."""
import torch
import torch.nn as nn

from .. import ops
from .linear import Linear


class TiedLinear(nn.Module):
    """tell/modules/linear.py:37-50 - holds the shared Parameter as `.weight`."""

    def __init__(self, weight, transpose=False):
        super().__init__()
        self.weight = weight
        self.transpose = transpose


class TiedHeadModule(nn.Module):
    """tell/modules/softmax.py:11-40."""

    def __init__(self, weights, input_dim, n_classes):
        super().__init__()
        tied_emb, _ = weights
        self.num_words, emb_dim = tied_emb.shape
        assert emb_dim == input_dim
        self.word_proj = TiedLinear(tied_emb)
        self.n_classes = n_classes
        self.class_proj = Linear(input_dim, n_classes, bias=False)
        self.out_dim = self.num_words + n_classes
        self.register_buffer('_float_tensor', torch.zeros(1))


class AdaptiveSoftmax(nn.Module):
    """Adaptive softmax tied to the adaptive input embedding (tie_adaptive_weights=True,
    tie_adaptive_proj=False, factor 1, dropout 0 - config.yaml:66-72)."""

    def __init__(self, vocab_size, input_dim, cutoff, dropout=0, factor=1., adaptive_inputs=None, tie_proj=False):
        super().__init__()
        if adaptive_inputs is None or tie_proj or dropout:
            raise NotImplementedError('only the tied-embedding configuration of the expt/ configs is implemented')
        cutoff = list(cutoff)
        if not cutoff or vocab_size > cutoff[-1]:
            cutoff.append(vocab_size)
        assert vocab_size == cutoff[-1]
        self.vocab_size, self.cutoff, self.input_dim = vocab_size, cutoff, input_dim
        n_tails = len(cutoff) - 1
        self.head = TiedHeadModule(adaptive_inputs.weights_for_band(0), input_dim, n_tails)
        self.tail = nn.ModuleList()
        for i in range(n_tails):
            emb, proj = adaptive_inputs.weights_for_band(i + 1)
            self.tail.append(nn.Sequential(Linear(input_dim, proj.shape[1], bias=False), nn.Dropout(0.0),
                                           TiedLinear(emb)))
        self.register_buffer('version', torch.LongTensor([1]))

    def _tails(self):
        out = []
        for t in self.tail:
            out += [t[0].weight, t[2].weight]
        return out

    def loss(self, x, target, padding_idx):
        """-> (loss_sum in nats [1], sample_size [1] int32), both on the device."""
        if x.dim() == 3 and not x.is_contiguous() and x.transpose(0, 1).is_contiguous():
            # the decoder hands its T x B x C buffer over as a [B,T,C] view: the summed loss does not care about the row
            # order, so the (tiny) target is transposed instead of the activations (and of their gradient)
            x, target = x.transpose(0, 1), target.t().contiguous()
        return ops.adaptive_loss(x, target, self.cutoff, padding_idx, self.head.word_proj.weight,
                                 self.head.class_proj.weight, self._tails())

    def get_log_prob(self, X, target=None):
        assert target is None
        B, T, E = X.shape
        _, _, full = ops.adaptive_log_probs(ops.as2dc(X), self.cutoff, self.head.word_proj.weight,
                                            self.head.class_proj.weight, self._tails(), want_full=True)
        return full.view(B, T, self.vocab_size)

    def topk(self, X, k):
        """The k best (token, log-prob) of every position, best first, fused like `greedy` (beam search)."""
        B, T, E = X.shape
        tok, lp, _ = ops.adaptive_log_probs(ops.as2dc(X), self.cutoff, self.head.word_proj.weight,
                                            self.head.class_proj.weight, self._tails(), topk=k)
        return tok.view(B, T, k), lp.view(B, T, k)

    def greedy(self, X):
        """Fused arg-max over the full vocabulary (get_log_prob + topk(1),
        transformer_faces_objects.py:443-464) without materialising [N, vocab]."""
        B, T, E = X.shape
        tok, lp, _ = ops.adaptive_log_probs(ops.as2dc(X), self.cutoff, self.head.word_proj.weight,
                                            self.head.class_proj.weight, self._tails())
        return tok.view(B, T), lp.view(B, T)
