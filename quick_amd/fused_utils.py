"""Packed-space concatenation of QUICK layers along the output dimension.

Counterpart of ``QUICK_cat`` / ``fuse_qkv_quick`` (quick/awq/utils/fused_utils.py:97-159), generalised to
layers of *different* widths (GQA: q 4096 / k 1024 / v 1024), which the reference rejects with
``ValueError("All input layers must have the same shape")`` (fused_utils.py:138-142) although its packed
format concatenates cleanly whenever every width is a multiple of 128: per (k-tile, quarter) segment the
128-channel blocks of the layers simply follow each other (closed form in SURVEY.md section 8(a)).
"""
from typing import Optional, Tuple

import torch

from .linear import WQLinear_QUICK


def QUICK_cat(*input_layers: torch.Tensor, options: str, reshape_dims: Optional[Tuple[int, int]] = None) -> torch.Tensor:
    """Concatenate reference-order packed tensors along N.  ``options`` in {'qweight', 'qzeros', 'scales'}.

    Same call signature as the reference.  ``reshape_dims``, when given, is applied to every input as in the
    reference (and therefore requires equal shapes); by default each input [H, W] is viewed as
    qweight -> (H/2, 2W), qzeros / scales -> (4H, W/4), which is what lets unequal widths through.
    """
    if len(input_layers) < 2:
        raise ValueError("At least two input layers are required")
    H = input_layers[0].shape[0]
    for layer in input_layers[1:]:
        if layer.shape[0] != H:
            raise ValueError("All input layers must have the same number of rows")
    if options not in ("qweight", "qzeros", "scales"):
        raise ValueError("Unknown options provided or invalid reshape dimensions")
    gran = {"qweight": 64, "qzeros": 32, "scales": 256}[options]      # columns per 128 output channels
    for layer in input_layers:
        if layer.shape[1] % gran != 0:
            raise ValueError("every layer must hold a multiple of 128 output channels")
    if reshape_dims is not None:
        parts = [layer.reshape(*reshape_dims) for layer in input_layers]
    elif options == "qweight":
        parts = [layer.reshape(H // 2, layer.shape[1] * 2) for layer in input_layers]
    else:
        parts = [layer.reshape(H * 4, layer.shape[1] // 4) for layer in input_layers]
    return torch.cat(parts, dim=1).reshape(H, -1)


def fuse_qkv_quick(module, q_proj, k_proj, v_proj):
    """One WQLinear_QUICK computing [q | k | v] (fused_utils.py:97-117); accepts unequal widths."""
    dev = q_proj.qweight.device if module is None else next(iter(module.state_dict().values())).device
    qkv_layer = WQLinear_QUICK(
        q_proj.w_bit, q_proj.group_size, q_proj.in_features,
        q_proj.out_features + k_proj.out_features + v_proj.out_features,
        q_proj.bias is not None, dev, q_proj.k_split_1, q_proj.k_split_2)
    packed = [p.reference_order() for p in (q_proj, k_proj, v_proj)]
    qkv_layer._set_packed(
        QUICK_cat(*[p[0] for p in packed], options="qweight").to(dev),
        QUICK_cat(*[p[1] for p in packed], options="scales").to(dev),
        QUICK_cat(*[p[2] for p in packed], options="qzeros").to(dev),
        prepared=False)
    qkv_layer.bias = torch.cat([q_proj.bias, k_proj.bias, v_proj.bias], dim=0) if q_proj.bias is not None else None
    return qkv_layer
