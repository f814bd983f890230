"""Tensor-parallel sharding of an ARC-NVFP4 linear (new design; the reference has no distributed path,
SURVEY.md 2.3 / 8-e).  One process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over xGMI).

Column-parallel (q/k/v/gate/up): split N.  Rank r owns rows [r*N/p, (r+1)*N/p) of the packed weight and
the matching 128-row scale tiles; no collective on the data path (optionally an all-gather of the output).

Row-parallel (o_proj/down_proj): split the AUGMENTED K axis in units of 64 elements (= one scale-factor
atom = 32 packed bytes), so packed bytes and swizzled scale bytes slice cleanly and a primary/residual
pair is never separated from its scale.  Every rank contracts its K slice with the same alpha into fp32
partials and the partials are summed with ONE all-reduce.

Column -> row hand-off (the MLP's gate|up -> down, attention's q|k|v -> o): after a column-parallel linear every rank holds
only ITS columns of the activation, and the following row-parallel linear quantises that activation with ONE per-tensor
scale (max|x| / 2688, model/qLlamaLayer.py:73-77) and a reorder_index over the WHOLE row.  Two ways, both here:

  A  ``handoff_gather``: all-gather the bf16 column blocks ((p-1)/p * M * N_inter * 2 B per rank), quantise the full row
     replicated, slice the packed bytes to the rank's K range (``shard_k``).  Bit-identical to the unsharded layer, any
     reorder_index.  Decode (M = 4, Llama-3-70B, N_inter = 28672): 201 KB per rank and layer -- latency-bound, fine;
     prefill (M = 4096): 206 MB -- as much as the weights: use B.
  B  ``handoff_local_scale``: every rank quantises only its own columns with a SHARD-LOCAL reorder_index / select_num
     (calibrated per shard; the weight shard is quantised with the same local index), and only the per-tensor scale is
     global: one all-reduce(MAX) of a 4-byte abs-max word.  No activation moves.  Each rank's bytes equal the unsharded
     quantiser run on that rank's column slice with the global scale; the layer equals the unsharded one whose
     reorder_index is the concatenation of the local ones.

All slicing helpers are pure tensor ops (device-agnostic), so the CPU tests exercise them with gloo.
"""
from __future__ import annotations

from typing import Tuple

import torch

ATOM = 64            # K elements per scale-factor atom column
TILE_ROWS = 128      # rows per scale-factor tile


def _sf_tiles(SF: torch.Tensor, rows: int, K: int) -> torch.Tensor:
    """View the swizzled scale buffer as [row_tiles, K atoms, 512 bytes] (only whole tiles)."""
    atoms = K // ATOM
    tiles = SF.numel() // (atoms * 512)
    need = (rows + TILE_ROWS - 1) // TILE_ROWS
    if tiles < need:
        raise ValueError(f"scale buffer holds {tiles} row tiles, {need} needed")
    return SF[: tiles * atoms * 512].view(tiles, atoms, 512)


def k_slices(K: int, world: int):
    """Balanced split of K (a multiple of 64) into `world` contiguous ranges on 64-element boundaries."""
    if K % ATOM:
        raise ValueError("K must be a multiple of 64")
    atoms = K // ATOM
    base, extra = divmod(atoms, world)
    out, a = [], 0
    for r in range(world):
        n = base + (1 if r < extra else 0)
        out.append((a * ATOM, (a + n) * ATOM))
        a += n
    return out


def shard_k(Q: torch.Tensor, SF: torch.Tensor, k0: int, k1: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Slice a packed operand [rows, K/2] + swizzled scales to the K range [k0, k1) (multiples of 64)."""
    rows, K = Q.shape[0], Q.shape[1] * 2
    if k0 % ATOM or k1 % ATOM or not (0 <= k0 <= k1 <= K):
        raise ValueError("K range must lie on 64-element boundaries")
    q = Q[:, k0 // 2: k1 // 2].contiguous()
    sf = _sf_tiles(SF, rows, K)[:, k0 // ATOM: k1 // ATOM].contiguous().view(-1)
    return q, sf


def n_slices(N: int, world: int):
    """Split N into `world` ranges aligned to 128 rows (scale tiles stay self-contained)."""
    tiles = (N + TILE_ROWS - 1) // TILE_ROWS
    base, extra = divmod(tiles, world)
    out, t = [], 0
    for r in range(world):
        n = base + (1 if r < extra else 0)
        out.append((min(t * TILE_ROWS, N), min((t + n) * TILE_ROWS, N)))
        t += n
    return out


def shard_n(Q: torch.Tensor, SF: torch.Tensor, n0: int, n1: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Slice a packed weight [N, K/2] + swizzled scales to rows [n0, n1) (n0 a multiple of 128)."""
    N, K = Q.shape[0], Q.shape[1] * 2
    if n0 % TILE_ROWS or not (0 <= n0 <= n1 <= N):
        raise ValueError("row range must start on a 128-row boundary")
    q = Q[n0:n1].contiguous()
    t0, t1 = n0 // TILE_ROWS, (n1 + TILE_ROWS - 1) // TILE_ROWS
    tiles = _sf_tiles(SF, N, K)[t0:t1]
    # keep the reference's "+1 spare tile" allocation rule so the shard is itself a valid operand
    spare = torch.zeros((1,) + tuple(tiles.shape[1:]), dtype=SF.dtype, device=SF.device)
    sf = torch.cat([tiles, spare], dim=0).contiguous().view(-1)
    return q, sf


class ColumnParallelARCLinear:
    """y[:, n0:n1] = x . W[n0:n1]^T on this rank; `gather_output` all-gathers the column blocks."""

    def __init__(self, QW, SFW, scale_w, rank: int, world: int, bias=None, group=None):
        self.rank, self.world, self.group = rank, world, group
        self.N = QW.shape[0]
        self.ranges = n_slices(self.N, world)
        n0, n1 = self.ranges[rank]
        self.W, self.SFW = shard_n(QW, SFW, n0, n1)
        self.scale_w = scale_w
        self.bias = None if bias is None else bias[n0:n1].contiguous()

    def forward(self, qx, sfx, scale_x, gather_output: bool = False):
        from . import agemm
        y = agemm.matmul(qx, self.W, sfx, self.SFW, scale_x * self.scale_w)
        if self.bias is not None:
            y = y + self.bias
        if not gather_output or self.world == 1:
            return y
        import torch.distributed as dist
        return all_gather_columns(y, [b - a for a, b in self.ranges], group=self.group)


def all_gather_columns(y: torch.Tensor, widths, group=None) -> torch.Tensor:
    """All-gather column blocks of possibly different widths (collectives need equal shapes: pad to the widest)."""
    import torch.distributed as dist
    world, wmax = len(widths), max(widths)
    if y.shape[1] < wmax:
        y = torch.nn.functional.pad(y, (0, wmax - y.shape[1]))
    rows = y.shape[0]
    out = torch.empty((world * rows, wmax), dtype=y.dtype, device=y.device)     # ranks concatenated along dim 0
    dist.all_gather_into_tensor(out, y.contiguous(), group=group)
    out = out.view(world, rows, wmax)
    return torch.cat([out[r, :, : widths[r]] for r in range(world)], dim=1)


class RowParallelARCLinear:
    """y = all_reduce_sum_r( x[:, Kr] . W[:, Kr]^T ) with Kr this rank's slice of the augmented K axis."""

    def __init__(self, QW, SFW, scale_w, rank: int, world: int, bias=None, group=None):
        self.rank, self.world, self.group = rank, world, group
        self.K = QW.shape[1] * 2
        self.k0, self.k1 = k_slices(self.K, world)[rank]
        self.W, self.SFW = shard_k(QW, SFW, self.k0, self.k1)
        self.scale_w = scale_w
        self.bias = bias

    def shard_activation(self, qx, sfx):
        """The activation is quantised on the FULL row (its per-16 scales and residual channels do not
        depend on the split) and then sliced to this rank's K range."""
        return shard_k(qx, sfx, self.k0, self.k1)

    def forward(self, qx, sfx, scale_x, out_dtype=torch.bfloat16):
        from . import agemm
        a, sfa = self.shard_activation(qx, sfx)
        part = agemm.matmul(a, self.W, sfa, self.SFW, scale_x * self.scale_w, out_dtype=torch.float32)
        if self.world > 1:
            import torch.distributed as dist
            dist.all_reduce(part, op=dist.ReduceOp.SUM, group=self.group)      # RCCL over xGMI
        if self.bias is not None:
            part = part + self.bias.float()
        return part.to(out_dtype)


# ---- column -> row hand-off ------------------------------------------------------------------------------------------------
def absmax_word(x: torch.Tensor) -> torch.Tensor:
    """max |x| of a bf16 tensor as ONE int32 word of bf16 magnitude bits -- the abs-max slot format of
    ``agemm.reorder_quantize_x_dynamic(..., absmax_slots=...)``; integer max == float max for magnitudes."""
    if x.dtype != torch.bfloat16:
        raise ValueError("absmax_word: bf16 tensor expected")
    return (x.contiguous().view(torch.int16).to(torch.int32) & 0x7FFF).max().reshape(1)


def handoff_gather(y_local: torch.Tensor, widths, group=None) -> torch.Tensor:
    """Hand-off A: all-gather the column blocks of a column-parallel output; the caller quantises the full row (replicated)
    and slices its K range with ``shard_k``.  Moves (p-1) * M * max(widths) * 2 bytes per rank."""
    return all_gather_columns(y_local, widths, group=group)


def handoff_local_scale(y_local: torch.Tensor, group=None) -> torch.Tensor:
    """Hand-off B: the GLOBAL per-tensor abs-max word from the rank-local ones: one all-reduce(MAX) of 4 bytes.  Feed it to
    ``agemm.reorder_quantize_x_dynamic(y_local, local_reorder_index, KE_local, absmax_slots=word)``: the rank quantises only
    its own columns, with the scale the unsharded layer would use."""
    import torch.distributed as dist
    word = absmax_word(y_local)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(word, op=dist.ReduceOp.MAX, group=group)
    return word


def handoff_bytes_per_rank(M: int, n_inter: int, world: int, mode: str) -> int:
    """Bytes a rank sends per column -> row hand-off (documentation / bench)."""
    if mode == "gather":
        return (world - 1) * M * ((n_inter + world - 1) // world) * 2
    if mode == "local_scale":
        return 4 * (world - 1)
    raise ValueError(mode)
