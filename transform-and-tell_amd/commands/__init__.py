from .evaluate import evaluate, evaluate_from_file, write_to_json  # noqa: F401
from .compute_metrics import compute_metrics  # noqa: F401
