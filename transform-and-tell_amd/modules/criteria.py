"""tell/modules/criteria/{base,adaptive_loss}.py on the MI355X path."""
import torch.nn as nn

from ..common.registrable import Registrable


class Criterion(nn.Module, Registrable):
    pass


@Criterion.register('adaptive_loss')
class AdaptiveLoss(Criterion):
    """Sum over clusters of cross_entropy(ignore_index=padding_idx, reduction='sum')
    (adaptive_loss.py:27-73).  Returns DEVICE scalars (loss in nats, sample_size) so the
    training step never synchronises with the host."""

    def __init__(self, padding_idx=1):
        super().__init__()
        self.padding_idx = padding_idx

    def forward(self, adaptive_softmax, net_output, decoder_target, reduction='sum'):
        assert reduction == 'sum'
        return adaptive_softmax.loss(net_output[0], decoder_target, self.padding_idx)
