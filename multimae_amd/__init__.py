"""MI355X-native MultiMAE pretraining step: hand-written gfx950 kernels (libmmae_hip.so, C ABI in include/mmae.h)
behind the module API of thinh-re/multimae (MultiMAE, PatchedInputAdapter, SpatialOutputAdapter, masked losses)."""
from .criterion import MaskedCrossEntropyLoss, MaskedL1Loss, MaskedMSELoss
from .input_adapters import PatchedInputAdapter, SemSegInputAdapter
from .multimae import MultiMAE, MultiViT, create_model, multivit_base, multivit_large, pretrain_multimae_base, pretrain_multimae_large
from .output_adapters import SpatialOutputAdapter

__all__ = ["MultiMAE", "MultiViT", "multivit_base", "multivit_large", "PatchedInputAdapter", "SemSegInputAdapter", "SpatialOutputAdapter", "MaskedMSELoss", "MaskedL1Loss",
           "MaskedCrossEntropyLoss",
           "create_model", "pretrain_multimae_base", "pretrain_multimae_large"]
