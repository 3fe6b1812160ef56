"""Host-side mirrors of tell/modules (same class names, constructor arguments and
state_dict keys); every forward/backward is HIP kernels through ops.py."""
from .attention import MultiHeadAttention  # noqa: F401
from .convolutions import DynamicConv1dTBC, LightweightConv1dTBC  # noqa: F401
from .criteria import AdaptiveLoss, Criterion  # noqa: F401
from .linear import GehringLinear  # noqa: F401
from .lstm import AttentionLayer, LSTMCell  # noqa: F401
from .softmax import AdaptiveSoftmax  # noqa: F401
from .token_embedders import (AdaptiveEmbedding, SinusoidalPositionalEmbedding,  # noqa: F401
                              SumTextFieldEmbedder, make_positions)
