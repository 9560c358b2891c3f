"""`from utils.optim_factory import create_optimizer` (run_pretraining_multimae.py:53 of the reference; the module itself is
not in the reference tree): the optimizer factory in the reference's call shape, create_optimizer(args, {"model": model})."""
from .utils import create_optimizer  # noqa: F401
