from .evaluate import evaluate, evaluate_from_file, write_to_json  # noqa: F401
