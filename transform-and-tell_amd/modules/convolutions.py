"""tell/modules/convolutions/dynamic.py:25-361 and lightweight.py:83-240 on the MI355X path."""
import torch
import torch.nn as nn

from .. import ops
from .linear import Linear

_INSTANCES = [0]


class DynamicConv1dTBC(nn.Module):
    """Dynamic lightweight convolution, T x B x C, as the decoders configure it
    (weight_softmax=True, padding_l=K-1 i.e. causal, no in_proj / renorm / conv bias)."""

    def __init__(self, input_size, kernel_size=1, padding_l=None, num_heads=1, weight_dropout=0.,
                 weight_softmax=True, renorm_padding=False, bias=False, conv_bias=False,
                 query_size=None, in_proj=False):
        super().__init__()
        if not weight_softmax or renorm_padding or conv_bias or in_proj or query_size is not None \
                or (padding_l is not None and padding_l != kernel_size - 1):
            raise NotImplementedError('only the causal weight_softmax configuration used by the '
                                      'expt/ decoders is implemented on the HIP path')
        self.input_size, self.kernel_size, self.num_heads = input_size, kernel_size, num_heads
        self.padding_l = kernel_size - 1
        self.weight_dropout = weight_dropout
        self.weight_linear = Linear(input_size, num_heads * kernel_size, bias=bias)
        _INSTANCES[0] += 1
        self._state_key = 'DynamicConv1dTBC.%d.input_buffer' % _INSTANCES[0]   # tell/utils/state.py:21-34

    def forward(self, X, incremental_state=None, query=None, unfold=None):
        assert query is None
        n_hist = 0
        if incremental_state is not None and incremental_state.get('_static'):
            # fixed-shape variant for the captured decode step: the buffer always holds K-1 rows, zero before the
            # caption starts.  Identical results: the tap softmax runs over all K taps and taps that reach before
            # the start multiply zero rows, which is what the reference's narrowing does (dynamic.py:306-311).
            prev = incremental_state[self._state_key]
            n_hist = prev.shape[0]
            if incremental_state.get('_ring'):
                # one token: tap logits of the new row (matrix cores), tap softmax, K-tap sum over the ring of past rows +
                # the row, and the row's store into the ring as ONE launch (tell_dynconv_step, the kernel of the fused decode
                # step) - not the K-row concatenation, a tap-logit GEMM over all K rows and the training kernel for K outputs
                # of which one is kept
                if not self._step_kernel_usable(X, prev):
                    raise RuntimeError('ring-buffer decode state with an input the step kernel does not take')
                M, C = X.shape[1], X.shape[2]
                y = torch.empty(1, M, C, dtype=X.dtype, device=X.device)
                ops.call('tell_dynconv_step', X.reshape(M, C), prev, ops.weight(self.weight_linear.weight), y, M, C,
                         self.num_heads, self.kernel_size, int(incremental_state['_t_cur']), incremental_state.get('_back'))
                return y
            X = torch.cat([prev, X], dim=0)
            if n_hist:
                prev.copy_(X[-n_hist:])
        elif incremental_state is not None:                          # dynamic.py:94-99
            prev = incremental_state.get(self._state_key)
            if prev is not None:
                n_hist = prev.shape[0]
                X = torch.cat([prev, X], dim=0)
            incremental_state[self._state_key] = X[-self.kernel_size + 1:] if self.kernel_size > 1 else X[:0]
        logits = self._logits(X)
        out = ops.dynamic_conv(X, logits, self.num_heads, self.kernel_size, self.weight_dropout, self.training)
        return out[n_hist:] if n_hist else out                       # dynamic.py:115-116

    def _logits(self, X):
        return self.weight_linear(X)

    def ring_usable(self):
        """tell_dynconv_step takes this module (its limits, mirrored): taps predicted from the input without a bias, 64-wide
        heads, 2 <= K <= 32, C = 512 / 1024 / 2048, bf16 compute."""
        C, H, K = self.input_size, self.num_heads, self.kernel_size
        return (type(self) is DynamicConv1dTBC and 2 <= K <= 32 and C == H * 64 and C in (512, 1024, 2048) and
                self.weight_linear.bias is None and ops.rt.compute_dtype() == torch.bfloat16)

    def _step_kernel_usable(self, X, prev):
        return (self.ring_usable() and not self.training and X.is_cuda and X.dtype == torch.bfloat16 and X.shape[0] == 1 and
                X.is_contiguous() and prev.is_contiguous() and prev.dtype == torch.bfloat16 and
                prev.shape[0] == self.kernel_size and prev.shape[1] == X.shape[1])

    def reorder_incremental_state(self, incremental_state, new_order):
        buf = incremental_state.get(self._state_key)
        if buf is not None:
            incremental_state[self._state_key] = buf.index_select(1, new_order)


class LightweightConv1dTBC(DynamicConv1dTBC):
    """tell/modules/convolutions/lightweight.py:83-240 as `decoder_conv_type: lightweight` builds it
    (decoder_faces_objects.py:199-203): one learned tap vector per head instead of taps predicted from the input.
    Same causal K-tap kernels; only the source of the tap logits differs (`ops.static_taps`)."""

    def __init__(self, input_size, kernel_size=1, padding_l=None, num_heads=1, weight_dropout=0.,
                 weight_softmax=True, bias=False):
        nn.Module.__init__(self)
        if not weight_softmax or bias or (padding_l is not None and padding_l != kernel_size - 1):
            raise NotImplementedError('only the causal weight_softmax configuration the decoders build is implemented')
        self.input_size, self.kernel_size, self.num_heads = input_size, kernel_size, num_heads
        self.padding_l = kernel_size - 1
        self.weight_dropout = weight_dropout
        self.weight = nn.Parameter(torch.empty(num_heads, 1, kernel_size))
        nn.init.xavier_uniform_(self.weight)                                      # lightweight.py:127
        _INSTANCES[0] += 1
        self._state_key = 'LightweightConv1dTBC.%d.input_buffer' % _INSTANCES[0]

    def _logits(self, X):
        return ops.static_taps(self.weight, X.shape[0], X.shape[1])
