"""Calibration artefacts either side of the hot path.

The reference's offline calibration writes three ``torch.save``d dictionaries under ``./saved`` and its
runtime reads them back (reorder_indices.py:88-99 writer, model/main.py:110-122 reader):

    {model}_reorder_index_{dataset}_{metric}.pt   name -> 1-D integer permutation of the K channels
    {model}_select_num_{dataset}_{metric}.pt      name -> KE (multiple of 64, 0 <= KE <= K)
    {model}_act_scales_{dataset}_{metric}.pt      name -> 1-D fp32 per-channel statistic
    {model}_average_bits_{dataset}_{metric}.pt    name -> float (report only)

with ``name = 'layers.{i}.{self_attn|mlp}.{proj}.input'`` (utilize.py:136, model/qLlamaLayer.py:237).

This module reads and writes exactly those files (so artefacts calibrated with the reference load here and
vice versa), restates the three pure-tensor rules that produce them, and adds what the reference lacks: a
cache of the QUANTISED weights, so that a served model does not re-quantise every linear at start-up.

Differences from the reference, deliberate:
  * files are read with ``torch.load(..., weights_only=True)`` -- they only hold tensors and numbers -- where
    the reference unpickles with ``weights_only=False`` (model/main.py:120-122);
  * everything loaded is validated (permutation, KE rule) before it can reach a kernel, because the
    kernels index with it unchecked (the reference would read out of bounds on a corrupt file).
The model-running part of calibration (hooks over a HF model and a dataset, utilize.py:80-252,386-500) is
outside the hot path and not rebuilt.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass, field
from typing import Dict, Iterable, Optional

import torch

NAME_TEMPLATE = "layers.{}.{}.{}.input"
KINDS = ("reorder_index", "select_num", "act_scales", "average_bits")


def artifact_path(kind: str, model_name: str, dataset: str, metric: str, root: str = "./saved") -> str:
    """reorder_indices.py:88-90 / model/main.py:112-114: the file name of one artefact."""
    if kind not in KINDS:
        raise ValueError(f"unknown artefact kind {kind!r}; one of {KINDS}")
    return os.path.join(root, f"{model_name.lower()}_{kind}_{dataset.lower()}_{metric}.pt")


def layer_input_name(layer: int, block: str, proj: str) -> str:
    """'layers.{i}.{block}.{proj}.input' (utilize.py:136)."""
    return NAME_TEMPLATE.format(layer, block, proj)


# ---------------------------------------------------------------------------------------------------
# the three tensor rules of calibration
# ---------------------------------------------------------------------------------------------------
def channel_stat(x: torch.Tensor, running: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Per-channel max|x| over all tokens, folded into ``running`` by max (utilize.py:106-116, default metric)."""
    s = x.reshape(-1, x.shape[-1]).abs().amax(dim=0).float().cpu()
    return s if running is None else torch.maximum(running, s)


def reorder_index_from_stat(stat: torch.Tensor) -> torch.Tensor:
    """Ascending sort of the per-channel statistic, so the outlier channels come LAST (utilize.py:39-45)."""
    if stat.dim() != 1:
        raise ValueError("the channel statistic must be 1-dimensional")
    return torch.sort(stat, descending=False)[1]


def select_num_from_samples(x: torch.Tensor, reorder_index: torch.Tensor):
    """(KE, average_bits) of one linear from sample activations (utilize.py:465-478).

    Counts the entries above an eighth of their token's (signed) maximum -- the reference compares the raw
    values, not magnitudes -- and rounds the implied channel count up to the 64-element K atom."""
    keys = x.reshape(-1, x.shape[-1])
    k = keys.shape[1]
    keys = keys[:, reorder_index.to(torch.int64)]
    threshold = keys.max(dim=-1, keepdim=True)[0] * 0.125
    ratio = float((keys > threshold).sum()) / keys.numel()
    ke = int(math.ceil(k * ratio / 64) * 64)
    return ke, 4.5 * (k + ke) / k


# ---------------------------------------------------------------------------------------------------
# validation
# ---------------------------------------------------------------------------------------------------
def check_reorder_index(name: str, idx: torch.Tensor, k: Optional[int] = None) -> None:
    if not torch.is_tensor(idx) or idx.dim() != 1 or idx.dtype.is_floating_point or idx.dtype == torch.bool:
        raise ValueError(f"{name}: the reorder index must be a 1-D integer tensor")
    n = idx.numel()
    if k is not None and n != k:
        raise ValueError(f"{name}: the reorder index has {n} entries, the layer has {k} input channels")
    if n > 32768:
        raise ValueError(f"{name}: {n} channels do not fit the kernels' int16 index")
    if n and not torch.equal(torch.sort(idx.to(torch.int64).cpu())[0], torch.arange(n)):
        raise ValueError(f"{name}: the reorder index is not a permutation of 0..{n - 1}")


def check_select_num(name: str, ke, k: int) -> int:
    ke_i = int(ke)
    if ke_i != ke or ke_i < 0 or ke_i % 64 or ke_i > k:
        raise ValueError(f"{name}: select_num={ke} must be a multiple of 64 in [0, {k}]")
    return ke_i


@dataclass
class Calibration:
    """The artefacts of one (model, dataset, metric) calibration, keyed by layer-input name."""
    reorder_index: Dict[str, torch.Tensor]
    select_num: Dict[str, int]
    act_scales: Dict[str, torch.Tensor] = field(default_factory=dict)
    average_bits: Dict[str, float] = field(default_factory=dict)

    def validate(self) -> "Calibration":
        for name, idx in self.reorder_index.items():
            check_reorder_index(name, idx)
        for name, ke in self.select_num.items():
            if name not in self.reorder_index:
                raise ValueError(f"{name}: select_num without a reorder index")
            self.select_num[name] = check_select_num(name, ke, self.reorder_index[name].numel())
        for name, s in self.act_scales.items():
            if name in self.reorder_index and s.numel() != self.reorder_index[name].numel():
                raise ValueError(f"{name}: act_scales and reorder index disagree on the channel count")
        return self

    def device_index(self, name: str, device) -> torch.Tensor:
        """The int16 device tensor the kernels take (model/model_utils.py:39-42, model/qLinearLayer.py:45)."""
        return self.reorder_index[name].to(device=device, dtype=torch.int16)

    def total_average_bits(self) -> float:
        """utilize.py:471-473,499: element-weighted bits per activation/weight element, 4.5*(K+KE)/K."""
        tot_k = sum(self.reorder_index[n].numel() for n in self.select_num)
        tot_b = sum(4.5 * (self.reorder_index[n].numel() + ke) for n, ke in self.select_num.items())
        return tot_b / tot_k if tot_k else 0.0


def calibration_from_stats(act_scales: Dict[str, torch.Tensor],
                           samples: Optional[Dict[str, torch.Tensor]] = None,
                           default_select_num: int = 0) -> Calibration:
    """Build the artefacts from per-channel statistics (and, for KE, sample activations) of every linear input.

    Names ending in '.output' are skipped like the reference does (utilize.py:52-59,466-467)."""
    reorder, select, bits = {}, {}, {}
    for name, stat in act_scales.items():
        if name.endswith(".output"):
            continue
        reorder[name] = reorder_index_from_stat(stat)
        k = stat.numel()
        if samples is not None and name in samples:
            select[name], bits[name] = select_num_from_samples(samples[name], reorder[name])
        else:
            select[name] = check_select_num(name, default_select_num, k)
            bits[name] = 4.5 * (k + select[name]) / k
    return Calibration(reorder, select, {n: s for n, s in act_scales.items() if not n.endswith(".output")}, bits).validate()


def save_calibration(cal: Calibration, model_name: str, dataset: str, metric: str, root: str = "./saved") -> Dict[str, str]:
    """Write the reference's files (reorder_indices.py:88-99; act_scales :55-63)."""
    cal.validate()
    os.makedirs(root, exist_ok=True)
    out = {}
    for kind in KINDS:
        obj = getattr(cal, kind)
        if kind in ("act_scales", "average_bits") and not obj:
            continue
        path = artifact_path(kind, model_name, dataset, metric, root)
        torch.save(dict(obj), path)
        out[kind] = path
    return out


def load_calibration(model_name: str, dataset: str, metric: str, root: str = "./saved",
                     require_act_scales: bool = False) -> Calibration:
    """Read the reference's files (model/main.py:110-122) without executing anything from them."""
    def read(kind, required):
        path = artifact_path(kind, model_name, dataset, metric, root)
        if not os.path.isfile(path):
            if required:
                raise FileNotFoundError(f"{kind} file not found: {path}")     # the reference asserts (main.py:117)
            return {}
        obj = torch.load(path, map_location="cpu", weights_only=True)
        if not isinstance(obj, dict):
            raise ValueError(f"{path}: expected a dict keyed by layer-input name")
        return obj
    return Calibration(read("reorder_index", True), read("select_num", True),
                       read("act_scales", require_act_scales), read("average_bits", False)).validate()


# ---------------------------------------------------------------------------------------------------
# quantised-weight cache (new: the reference re-quantises every weight at each start, qLinearLayer.py:45-66)
# ---------------------------------------------------------------------------------------------------
_CACHE_VERSION = 1


def save_quantized_weights(layers: Dict[str, "torch.nn.Module"], path: str) -> None:
    """Store (W, scale_w, scale, bias, KE, shape) of every QLinearLayer in one file of plain tensors."""
    blob = {"__version__": _CACHE_VERSION}
    for name, m in layers.items():
        blob[name] = {
            "W": m.W.cpu(), "scale_w": m.scale_w.cpu(), "scale": m.scale.cpu(),
            "bias": None if m.bias is None else m.bias.cpu(),
            "select_num": int(m.select_num), "in_features": int(m.in_features), "out_features": int(m.out_features),
        }
    torch.save(blob, path)


def load_quantized_weights(path: str, device="cuda") -> Dict[str, dict]:
    """Read a cache written by :func:`save_quantized_weights`; shapes are checked against the NVFP4 layout."""
    from . import agemm
    blob = torch.load(path, map_location="cpu", weights_only=True)
    if not isinstance(blob, dict) or blob.get("__version__") != _CACHE_VERSION:
        raise ValueError(f"{path}: not a quantised-weight cache of version {_CACHE_VERSION}")
    out = {}
    for name, e in blob.items():
        if name == "__version__":
            continue
        n, kq, ke = e["out_features"], e["in_features"], check_select_num(name, e["select_num"], e["in_features"])
        k = kq + ke
        if tuple(e["W"].shape) != (n, k // 2) or e["W"].dtype != torch.uint8:
            raise ValueError(f"{name}: packed weight has shape {tuple(e['W'].shape)}, expected {(n, k // 2)} u8")
        if e["scale_w"].numel() != agemm.sf_buffer_bytes(n, k) or e["scale_w"].dtype != torch.uint8:
            raise ValueError(f"{name}: scale buffer has {e['scale_w'].numel()} bytes, expected {agemm.sf_buffer_bytes(n, k)}")
        out[name] = {key: (v.to(device) if torch.is_tensor(v) else v) for key, v in e.items()}
    return out


def restore_qlinear(entry: dict):
    """A QLinearLayer rebuilt from one cache entry, with no quantisation pass."""
    from .qlinear import QLinearLayer
    m = QLinearLayer.__new__(QLinearLayer)
    torch.nn.Module.__init__(m)
    m.in_features, m.out_features = entry["in_features"], entry["out_features"]
    m.select_num, m.quant_type = entry["select_num"], "NVFP4"
    if entry["bias"] is not None:
        m.register_buffer("bias", entry["bias"])
    else:
        m.bias = None
    m.register_buffer("W", entry["W"])
    m.register_buffer("scale_w", entry["scale_w"])
    m.register_buffer("scale", entry["scale"])
    m.register_buffer("RW", None)            # the optional decode copy is rebuilt on demand (agemm.repack_w), not cached
    m.register_buffer("RSF", None)
    return m


def names_for_decoder(n_layers: int) -> Iterable[str]:
    """The layer-input names a Llama/Qwen decoder stack uses (model/qLlamaLayer.py:241-259,392-406)."""
    for i in range(n_layers):
        for blk, proj in (("self_attn", "q_proj"), ("self_attn", "k_proj"), ("self_attn", "v_proj"), ("self_attn", "o_proj"),
                          ("mlp", "gate_proj"), ("mlp", "up_proj"), ("mlp", "down_proj")):
            yield layer_input_name(i, blk, proj)
