from .optimizers import BertAdam, FlatParams, warmup_linear  # noqa: F401
from .trainer import CallbackApexTrainer, Trainer  # noqa: F401
