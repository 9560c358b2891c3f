"""Model construction helpers mirroring get_model of the reference's training script
(run_pretraining_multimae.py:57-140 there): domain table -> adapters -> MultiMAE."""
from functools import partial
from typing import Dict, Sequence

from .criterion import MaskedL1Loss, MaskedMSELoss
from .input_adapters import PatchedInputAdapter
from .multimae import create_model
from .output_adapters import SpatialOutputAdapter

DOMAIN_CONF = {
    "rgb": {"channels": 3, "stride_level": 1, "loss": MaskedMSELoss},
    "depth": {"channels": 1, "stride_level": 1, "loss": MaskedL1Loss},
}


def build_multimae(model: str = "pretrain_multimae_base", in_domains: Sequence[str] = ("rgb", "depth"),
                   out_domains: Sequence[str] = ("rgb", "depth"), patch_size: int = 16, image_size: int = 224,
                   decoder_dim: int = 256, decoder_depth: int = 2, decoder_num_heads: int = 8,
                   extra_norm_pix_loss: bool = True, num_global_tokens: int = 1, **model_kwargs):
    ia = {d: PatchedInputAdapter(num_channels=DOMAIN_CONF[d]["channels"], stride_level=DOMAIN_CONF[d]["stride_level"],
                                 patch_size_full=patch_size, image_size=image_size) for d in in_domains}

    def dec(d, task):
        return SpatialOutputAdapter(num_channels=DOMAIN_CONF[d]["channels"], stride_level=DOMAIN_CONF[d]["stride_level"],
                                    patch_size_full=patch_size, dim_tokens=decoder_dim, depth=decoder_depth,
                                    num_heads=decoder_num_heads, use_task_queries=True, task=task,
                                    context_tasks=list(in_domains), use_xattn=True, image_size=image_size)
    oa = {d: dec(d, d) for d in out_domains}
    if extra_norm_pix_loss:
        oa["norm_rgb"] = dec("rgb", "rgb")
    return create_model(model, input_adapters=ia, output_adapters=oa, num_global_tokens=num_global_tokens, **model_kwargs)


def build_losses(out_domains: Sequence[str] = ("rgb", "depth"), patch_size: int = 16, extra_norm_pix_loss: bool = True) -> Dict:
    """task -> (loss module, target domain), as in the reference's main() (run_pretraining_multimae.py:306-317 there)."""
    tab = {d: (DOMAIN_CONF[d]["loss"](patch_size=patch_size, stride=DOMAIN_CONF[d]["stride_level"]), d) for d in out_domains}
    if extra_norm_pix_loss:
        tab["norm_rgb"] = (MaskedMSELoss(patch_size=patch_size, stride=1, norm_pix=True), "rgb")
    return tab
