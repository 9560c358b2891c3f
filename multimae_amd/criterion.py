"""Masked losses with the reference's constructor / forward contract (multimae/criterion.py)."""
import torch
from torch import nn

from . import functional as Fn


class _MaskedPixelLoss(nn.Module):
    kind = 0

    def __init__(self, patch_size: int = 16, stride: int = 1, norm_pix=False):
        super().__init__()
        self.patch_size = patch_size
        self.stride = stride
        self.scale_factor = patch_size // stride
        self.norm_pix = norm_pix

    def forward(self, input, target, mask=None):
        """input [B,C,H,W] (fp32 or bf16), target fp32, mask [B, nh*nw] (1 = masked = contributes) or None.

        Mean over channels, masked mean over pixels per sample, mean over the samples that have masked tokens.
        One difference from the reference: when no token is masked at all it returns an fp32 zero (the reference
        returns an int64 zero) because the count stays on the device."""
        if mask is not None and mask.dtype != torch.int64:
            mask = mask.to(torch.int64)
        return Fn.MaskedLossFn.apply(input, target.float(), None if mask is None else mask.contiguous(), self.kind,
                                     bool(self.norm_pix), self.scale_factor)


class MaskedMSELoss(_MaskedPixelLoss):
    """Reference: multimae/criterion.py:68-130."""
    kind = 0


class MaskedL1Loss(_MaskedPixelLoss):
    """Reference: multimae/criterion.py:133-195."""
    kind = 1


class MaskedCrossEntropyLoss(nn.Module):
    """Reference: multimae/criterion.py:23-65.  input [B,C,H,W] logits (fp32 or bf16), target int64 [B,H,W],
    mask [B, (H/scale)*(W/scale)] with scale = patch_size // stride.  As with the pixel losses, an all-zero mask
    yields an fp32 zero (the reference returns an int64 zero)."""

    def __init__(self, patch_size: int = 16, stride: int = 1, label_smoothing: float = 0.0):
        super().__init__()
        self.patch_size = patch_size
        self.stride = stride
        self.scale_factor = patch_size // stride
        self.label_smoothing = label_smoothing

    def forward(self, input, target, mask=None):
        if mask is not None and mask.dtype != torch.int64:
            mask = mask.to(torch.int64)
        return Fn.MaskedCEFn.apply(input, target.to(torch.int64), None if mask is None else mask.contiguous(),
                                   self.scale_factor, float(self.label_smoothing))
