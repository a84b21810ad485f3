"""Round-to-nearest group quantisation and the hook that turns ``nn.Linear`` layers into ``WQLinear_QUICK``.

This is the part of the reference's quantizer that sits on the GEMM path: ``AwqQuantizer.pseudo_quantize_tensor``
(quick/awq/quantize/quantizer.py:46-72) and ``_apply_quant`` / ``pack`` (:141-174).  The AWQ scale / clip *search*
(:88-133, the calibration loop) is out of scope; a model whose weights went through that search
is packed by exactly the code below.  Pure torch, runs on CPU or GPU tensors.
"""
import torch
import torch.nn as nn

from .linear import WQLinear_QUICK


@torch.no_grad()
def pseudo_quantize_tensor(w, w_bit=4, group_size=128, get_scale_zp=False):
    """Asymmetric (zero-point) round-to-nearest per group of ``group_size`` input channels.

    ``w`` [out_features, in_features].  Returns the fake-quantised weight (same shape and dtype), and with
    ``get_scale_zp`` also ``scales`` and ``zeros`` as [out_features, in_features / group_size] -- the arguments
    ``WQLinear_QUICK.from_linear`` expects.  Same arithmetic, in the dtype of ``w``, as quantizer.py:53-64.
    """
    shape = w.shape
    if group_size > 0:
        if shape[-1] % group_size != 0:
            raise ValueError(f"in_features ({shape[-1]}) is not a multiple of the group size ({group_size})")
        w = w.reshape(-1, group_size)
    if w.dim() != 2:
        raise ValueError("expected a 2-D weight")
    qmax = 2 ** w_bit - 1
    hi, lo = w.amax(dim=1, keepdim=True), w.amin(dim=1, keepdim=True)
    scales = (hi - lo).clamp(min=1e-5) / qmax
    zeros = (-torch.round(lo / scales)).clamp_(0, qmax)
    if torch.isnan(scales).any() or torch.isnan(w).any():
        raise ValueError("NaN in the weight")
    w = (torch.clamp(torch.round(w / scales) + zeros, 0, qmax) - zeros) * scales
    w = w.reshape(shape)
    if get_scale_zp:
        return w, scales.view(shape[0], -1), zeros.view(shape[0], -1)
    return w


@torch.no_grad()
def quantize_linear(linear, w_bit=4, group_size=128):
    """``nn.Linear`` -> ``WQLinear_QUICK`` holding its round-to-nearest quantisation (the body of ``_apply_quant``,
    quantizer.py:148-174).  Stays on the device of ``linear``; the weight is quantised in fp16 as the reference does."""
    if w_bit != 4:
        raise NotImplementedError("Only 4-bit are supported for now.")
    lin = nn.Linear(linear.in_features, linear.out_features, linear.bias is not None, device=linear.weight.device,
                    dtype=torch.float16)
    wq, scales, zeros = pseudo_quantize_tensor(linear.weight.data.half(), w_bit, group_size, get_scale_zp=True)
    lin.weight.data = wq
    if linear.bias is not None:
        lin.bias.data = linear.bias.data.half()
    return WQLinear_QUICK.from_linear(lin, w_bit, group_size, False, scales, zeros)


@torch.no_grad()
def quantize_module_linears(module, w_bit=4, group_size=128, modules_to_not_convert=()):
    """Replace every ``nn.Linear`` below ``module`` whose name does not contain one of ``modules_to_not_convert`` (the
    reference's ``exclude_layers_to_not_quantize``) by its ``WQLinear_QUICK``; the ``pack()`` loop of quantizer.py:141-146.
    Returns the names replaced."""
    done = []
    for name, child in list(module.named_modules()):
        if not isinstance(child, nn.Linear) or any(key in name for key in modules_to_not_convert):
            continue
        parent = module
        *path, leaf = name.split(".")
        for part in path:
            parent = getattr(parent, part)
        setattr(parent, leaf, quantize_linear(child, w_bit, group_size))
        done.append(name)
    return done
