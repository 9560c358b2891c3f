"""Model construction helpers mirroring get_model of the reference's training script
(run_pretraining_multimae.py:57-140 there): domain table -> adapters -> MultiMAE."""
from functools import partial
from typing import Dict, Sequence

from .criterion import MaskedCrossEntropyLoss, MaskedL1Loss, MaskedMSELoss
from .input_adapters import PatchedInputAdapter, SemSegInputAdapter
from .multimae import create_model
from .output_adapters import SpatialOutputAdapter

DOMAIN_CONF = {
    "rgb": {"channels": 3, "stride_level": 1, "loss": MaskedMSELoss},
    "depth": {"channels": 1, "stride_level": 1, "loss": MaskedL1Loss},
    # class map at a quarter of the image resolution; 133 COCO panoptic classes, 64-wide class embedding
    # (run_pretraining_multimae.py:72-85 of the reference)
    "semseg": {"num_classes": 133, "dim_class_emb": 64, "stride_level": 4, "loss": MaskedCrossEntropyLoss},
}


def build_multimae(model: str = "pretrain_multimae_base", in_domains: Sequence[str] = ("rgb", "depth"),
                   out_domains: Sequence[str] = ("rgb", "depth"), patch_size: int = 16, image_size: int = 224,
                   decoder_dim: int = 256, decoder_depth: int = 2, decoder_num_heads: int = 8,
                   extra_norm_pix_loss: bool = True, num_global_tokens: int = 1, **model_kwargs):
    def inp(d):
        conf = DOMAIN_CONF[d]
        if "num_classes" in conf:
            return SemSegInputAdapter(num_classes=conf["num_classes"], dim_class_emb=conf["dim_class_emb"], interpolate_class_emb=False,
                                      stride_level=conf["stride_level"], patch_size_full=patch_size, image_size=image_size)
        return PatchedInputAdapter(num_channels=conf["channels"], stride_level=conf["stride_level"], patch_size_full=patch_size,
                                   image_size=image_size)
    ia = {d: inp(d) for d in in_domains}

    def dec(d, task):
        return SpatialOutputAdapter(num_channels=DOMAIN_CONF[d].get("channels", DOMAIN_CONF[d].get("num_classes")),
                                    stride_level=DOMAIN_CONF[d]["stride_level"],
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
