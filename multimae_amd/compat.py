"""Import aliases so that scripts written against the reference's package names pick up this implementation:

    import multimae_amd.compat; multimae_amd.compat.install()
    from multimae import MultiMAE                      # -> multimae_amd.MultiMAE
    from multimae.criterion import MaskedMSELoss       # -> multimae_amd.criterion.MaskedMSELoss
    import utils; from utils import create_model       # -> multimae_amd.utils (the step-side helpers of the reference's utils)
    from utils.optim_factory import create_optimizer   # -> multimae_amd.optim_factory.create_optimizer
"""
import sys

_ALIASES = {
    "multimae": "multimae_amd",
    "multimae.multimae": "multimae_amd.multimae",
    "multimae.multimae_utils": "multimae_amd.multimae_utils",
    "multimae.input_adapters": "multimae_amd.input_adapters",
    "multimae.output_adapters": "multimae_amd.output_adapters",
    "multimae.criterion": "multimae_amd.criterion",
    "utils": "multimae_amd.utils",
    "utils.optim_factory": "multimae_amd.optim_factory",
}


def install():
    """Register the aliases; returns the previous sys.modules entries (pass them to uninstall())."""
    import importlib
    previous = {}
    for alias, target in _ALIASES.items():
        previous[alias] = sys.modules.get(alias)
        sys.modules[alias] = importlib.import_module(target)
    return previous


def uninstall(previous):
    for alias, mod in previous.items():
        if mod is None:
            sys.modules.pop(alias, None)
        else:
            sys.modules[alias] = mod
