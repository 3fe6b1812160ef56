from .synthetic import SyntheticReader, synthetic_batch  # noqa: F401
