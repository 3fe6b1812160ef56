from .bleu import BleuScorer  # noqa: F401
from .cider import CiderScorer  # noqa: F401
from .rouge import Rouge  # noqa: F401
