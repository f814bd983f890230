"""Host-side mirror of the reference's quantised-linear operator.

Mirrors, with the same names, argument meaning and tuple protocol:
  * ``NVFP4_reorder_quantize_w``  model/qLinearLayer.py:25-28
  * ``NVFP4_reorder_quantize_x``  model/qLlamaLayer.py:73-77 == model/qQwenLayer.py:72-75
  * ``QLinearLayer``              model/qLinearLayer.py:30-78  (forward takes ``(qx, scale_x, scale, bsz, q_len)``)

so that the decoder-layer wrappers of the reference (qLlamaLayer.py / qQwenLayer.py) run unchanged on
top of ``arcquant_amd.agemm``.  What differs from the reference, deliberately:
  * no ``torch.cuda.synchronize()`` after every op and no ``.item()`` on the scale: the per-tensor scales
    stay on the device (0-dim fp32 tensors) and everything is ordered on torch's current stream;
  * only ``quant_type='NVFP4'`` is built (the other types are the reference's fake-quant study paths).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import agemm

FP8_MAX = 448.0
FP4_MAX = 6.0


def NVFP4_reorder_quantize_w(w: torch.Tensor, reorder_index: torch.Tensor, select_num: int):
    """(qw, scale_w, scale): per-tensor scale = max(w)/2688 -- the SIGNED max, as in the reference."""
    scale = torch.max(w).float() / (FP8_MAX * FP4_MAX)
    qw, scale_w = agemm.reorder_quantize_w((w / scale).contiguous(), reorder_index, select_num)
    return qw, scale_w, scale


def NVFP4_reorder_quantize_x(x: torch.Tensor, reorder_index: torch.Tensor, select_num: int):
    """(qx, scale_x, scale): per-tensor scale = max|x|/2688."""
    scale = torch.max(x.abs()).float() / (FP8_MAX * FP4_MAX)
    qx, scale_x = agemm.reorder_quantize_x((x / scale).contiguous(), reorder_index, select_num)
    return qx, scale_x, scale


def reorder_quantize_x(x, reorder_index, select_num, quant_type="NVFP4"):
    """model/qLlamaLayer.py:79-86 (NVFP4 branch)."""
    if quant_type != "NVFP4":
        raise NotImplementedError("only the NVFP4 path is part of the MI355X hot path")
    return NVFP4_reorder_quantize_x(x, reorder_index, select_num)


def find_qlinear_layers(module, name=""):
    """model/qLinearLayer.py:14-23."""
    if type(module) == QLinearLayer:
        return {name: module}
    res = {}
    for child_name, child in module.named_children():
        res.update(find_qlinear_layers(child, name=name + "." + child_name if name != "" else child_name))
    return res


class QLinearLayer(nn.Module):
    """Weight quantised once at construction; forward = ARC-NVFP4 GEMM (+ bias) on pre-quantised activations."""

    def __init__(self, originalLayer: nn.Linear, select_num, reorder_index, out_reorder_index=None, quant_type="NVFP4",
                 repack_for_decode: bool = False):
        """``repack_for_decode`` (extension, default off = the reference's behaviour and memory): additionally keep the
        weight in MFMA-operand-order tiles (``agemm.repack_w``) and use ``agemm.matmul_repacked`` for calls of at most 16
        tokens."""
        super().__init__()
        if quant_type != "NVFP4":
            raise NotImplementedError("only quant_type='NVFP4' is supported")
        self.in_features = originalLayer.in_features
        self.out_features = originalLayer.out_features
        if originalLayer.bias is not None:
            self.register_buffer("bias", originalLayer.bias.data)
        else:
            self.bias = None
        self.select_num = int(select_num)
        self.quant_type = quant_type
        dev = originalLayer.weight.device
        if dev.type != "cuda":
            raise RuntimeError("QLinearLayer: the weight must be on the GPU (quantisation runs there)")
        idx = reorder_index.to(device=dev, dtype=torch.int16)
        w = originalLayer.weight.data.to(torch.bfloat16)
        W, scale_w, scale = NVFP4_reorder_quantize_w(w, idx, self.select_num)
        self.register_buffer("W", W)
        self.register_buffer("scale_w", scale_w)
        self.register_buffer("scale", scale)
        RW, RSF = agemm.repack_w(W, scale_w) if repack_for_decode else (None, None)
        self.register_buffer("RW", RW)
        self.register_buffer("RSF", RSF)

    @torch.no_grad()
    def forward(self, x):
        qx, scale_x, scale, bsz, q_len = x
        # `y = matmul(...); y = y + bias` (model/qLinearLayer.py:74-76): the bias add runs in the GEMM epilogue with the same
        # two roundings (the bf16 product, then the bf16 sum), so the result is bit-identical to the two torch steps
        bias = self.bias if self.bias is None or self.bias.dtype == torch.bfloat16 else None
        if getattr(self, "RW", None) is not None and agemm.repacked_supported(qx.shape[0], self.out_features, qx.shape[1] * 2):
            y = agemm.matmul_repacked(qx, self.RW, scale_x, self.RSF, scale * self.scale, self.out_features, bias=bias)
        else:
            y = agemm.matmul(qx, self.W, scale_x, self.scale_w, scale * self.scale, bias=bias)
        if self.bias is not None and bias is None:
            y = y + self.bias
        return y.reshape(bsz, q_len, -1)
