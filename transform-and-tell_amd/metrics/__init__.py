from .bleu import BleuScorer  # noqa: F401
