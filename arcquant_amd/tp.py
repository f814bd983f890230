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


def concat_k(parts) -> Tuple[torch.Tensor, torch.Tensor]:
    """Inverse of ``shard_k``: join packed operands [(Q_r, SF_r), ...] of the same row count along K (each K_r a multiple of 64).
    Every K position contributes independently to the contraction, so the joined operand is the unsharded layer's."""
    rows = parts[0][0].shape[0]
    tiles = [_sf_tiles(sf, rows, q.shape[1] * 2) for q, sf in parts if q.shape[1]]
    nt = min(t.shape[0] for t in tiles)
    return (torch.cat([q for q, _ in parts], dim=1).contiguous(), torch.cat([t[:nt] for t in tiles], dim=1).contiguous().view(-1))


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


def gather_rows(Q: torch.Tensor, SF: torch.Tensor, ranges) -> Tuple[torch.Tensor, torch.Tensor]:
    """Concatenate several row ranges of a packed weight (each starting on a 128-row boundary and, except the last, a multiple of
    128 rows long) into one operand: the per-rank q|k|v shard of a GQA model is rows of q, of k and of v, not one contiguous
    range of the fused weight.  Scale tiles are 128 rows, so whole tiles are moved."""
    N, K = Q.shape[0], Q.shape[1] * 2
    qs, tiles = [], []
    T = _sf_tiles(SF, N, K)
    for i, (n0, n1) in enumerate(ranges):
        if n0 % TILE_ROWS or not (0 <= n0 <= n1 <= N) or (i + 1 < len(ranges) and (n1 - n0) % TILE_ROWS):
            raise ValueError("gather_rows: ranges must start on 128-row boundaries and (except the last) span whole 128-row tiles")
        qs.append(Q[n0:n1])
        tiles.append(T[n0 // TILE_ROWS: (n1 + TILE_ROWS - 1) // TILE_ROWS])
    spare = torch.zeros((1,) + tuple(T.shape[1:]), dtype=SF.dtype, device=SF.device)
    return torch.cat(qs, dim=0).contiguous(), torch.cat(tiles + [spare], dim=0).contiguous().view(-1)


def _ops(ops):
    if ops is not None:
        return ops
    from . import agemm
    return agemm


def _alpha(scale_x, scale_w):
    """(scale, scale_host) for ops.matmul: a device activation scale stays on the device, a float weight scale rides as scale_host."""
    if isinstance(scale_x, torch.Tensor) and isinstance(scale_w, torch.Tensor):
        return scale_x * scale_w, 1.0
    if isinstance(scale_x, torch.Tensor):
        return scale_x, float(scale_w)
    if isinstance(scale_w, torch.Tensor):
        return scale_w, float(scale_x)
    return float(scale_x) * float(scale_w), 1.0


class ColumnParallelARCLinear:
    """y[:, n0:n1] = x . W[n0:n1]^T (+ bias, in the GEMM epilogue) on this rank; `gather_output` all-gathers the column blocks.
    ``row_ranges``: this rank's rows as a list of (n0, n1) ranges instead of the balanced contiguous split (GQA q|k|v shards).
    ``repack_for_decode()`` adds the MFMA-operand-order copy of the shard; decode-sized calls then run the repacked / fused
    kernels (``forward_rmsnorm``, ``forward_rmsnorm_silu``: the activation quantiser as the GEMM's prologue, one launch)."""

    def __init__(self, QW, SFW, scale_w, rank: int, world: int, bias=None, group=None, row_ranges=None, ops=None):
        self.rank, self.world, self.group, self.ops = rank, world, group, _ops(ops)
        self.N = QW.shape[0]
        self.K = QW.shape[1] * 2
        self.ranges = n_slices(self.N, world)
        if row_ranges is None:
            n0, n1 = self.ranges[rank]
            self.W, self.SFW = shard_n(QW, SFW, n0, n1)
            self.bias = None if bias is None else bias[n0:n1].contiguous()
        else:
            self.W, self.SFW = gather_rows(QW, SFW, row_ranges)
            self.bias = None if bias is None else torch.cat([bias[a:b] for a, b in row_ranges]).contiguous()
        self.N_local = self.W.shape[0]
        self.scale_w = scale_w
        self.RW = self.RSF = None

    @classmethod
    def from_local(cls, QW_local, SFW_local, scale_w, rank: int, world: int, bias=None, group=None, ops=None):
        """A shard that was quantised by itself (its rows of the layer's weight, any per-tensor scale)."""
        self = cls.__new__(cls)
        self.rank, self.world, self.group, self.ops = rank, world, group, _ops(ops)
        self.W, self.SFW, self.bias, self.scale_w = QW_local, SFW_local, bias, scale_w
        self.N_local, self.K = QW_local.shape[0], QW_local.shape[1] * 2
        self.N = self.N_local * world
        self.ranges = [(r * self.N_local, (r + 1) * self.N_local) for r in range(world)]
        self.RW = self.RSF = None
        return self

    def repack_for_decode(self):
        self.RW, self.RSF = self.ops.repack_w(self.W, self.SFW)
        return self

    def forward(self, qx, sfx, scale_x, gather_output: bool = False):
        ops = self.ops
        scale, scale_host = _alpha(scale_x, self.scale_w)
        if self.RW is not None and ops.repacked_supported(qx.shape[0], self.N_local, self.K):
            y = ops.matmul_repacked(qx, self.RW, sfx, self.RSF, scale, self.N_local, bias=self.bias, scale_host=scale_host)
        else:
            y = ops.matmul(qx, self.W, sfx, self.SFW, scale, bias=self.bias, scale_host=scale_host)
        if not gather_output or self.world == 1:
            return y
        return all_gather_columns(y, [b - a for a, b in self.ranges], group=self.group)

    def forward_rmsnorm(self, X, Wn, eps, reorder_index, KE):
        """RMSNorm + quantise + this rank's column shard: ONE launch for decode-sized M (the replicated activation is quantised
        by every rank, as the unsharded layer does), two otherwise."""
        ops, M, KQ = self.ops, X.shape[0], X.shape[1]
        if self.RW is not None and ops.fused_supported(ops.SRC_RMSNORM, M, self.N_local, KQ, KE):
            return ops.rmsnorm_matmul_repacked(X, Wn, eps, reorder_index, KE, self.RW, self.RSF, self.scale_w, self.N_local, bias=self.bias)
        qx, sfx = ops.rmsnorm_quantize_x(X, Wn, eps, reorder_index, KE)
        return self.forward(qx, sfx, 1.0)

    def forward_rmsnorm_silu(self, X, Wn, eps, reorder_index, KE):
        """The MLP's first half on a gate|up shard whose ROWS INTERLEAVE gate and up: returns ``(act bf16 [M, N_local/2],
        absmax_word int32 [1])`` -- act = silu(gate) * up with torch's roundings, the word = max|act| of THIS rank's columns in
        the abs-max slot format (``handoff_local_scale`` reduces it over the ranks)."""
        ops, M, KQ = self.ops, X.shape[0], X.shape[1]
        if self.RW is not None and ops.fused_supported(ops.SRC_RMSNORM, M, self.N_local, KQ, KE):
            act, slots = ops.rmsnorm_matmul_repacked_silu(X, Wn, eps, reorder_index, KE, self.RW, self.RSF, self.scale_w, self.N_local, bias=self.bias)
            return act, slots.max().reshape(1)
        gu = self.forward_rmsnorm(X, Wn, eps, reorder_index, KE)
        act = (torch.nn.functional.silu(gu[:, 0::2]) * gu[:, 1::2]).contiguous()
        return act, absmax_word(act)


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


DECODE_M = 16        # up to here a row-parallel partial is reduced in fp32 by one all-reduce (<= 512 KB at N = 8192: latency-bound)


def all_reduce_sum(part: torch.Tensor, group=None, two_shot=None) -> torch.Tensor:
    """Sum the row-parallel partials over the ranks, in place.  Decode-sized fp32 partials: ONE all-reduce (latency-bound, the
    message is M x N x 4 B).  Larger bf16 partials: two-shot = reduce-scatter of row blocks + all-gather, every link carries
    (p-1)/p of the message once in each direction (xGMI is point-to-point: a ring all-reduce is per-link bound); needs
    M % world == 0 and a backend with reduce_scatter (RCCL), else one all-reduce."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if world == 1:
        return part
    M = part.shape[0]
    if two_shot is None:
        two_shot = M > DECODE_M and dist.get_backend(group) == "nccl"
    if two_shot and M % world == 0:
        scat = torch.empty((M // world,) + tuple(part.shape[1:]), dtype=part.dtype, device=part.device)
        dist.reduce_scatter_tensor(scat, part, op=dist.ReduceOp.SUM, group=group)
        dist.all_gather_into_tensor(part, scat, group=group)
    else:
        dist.all_reduce(part, op=dist.ReduceOp.SUM, group=group)
    return part


def finish_row_parallel(total: torch.Tensor, bias=None, residual=None) -> torch.Tensor:
    """The reduced partial sum -> the layer's bf16 output with the SAME roundings as the single-GPU epilogue
    (model/qLinearLayer.py:74-76): ``y = bf16(sum)``, then ``y = bf16(y + bias)``, then ``bf16(residual + y)``."""
    y = total.to(torch.bfloat16)
    if bias is not None:
        y = y + bias
    if residual is not None:
        y = residual + y
    return y


class RowParallelARCLinear:
    """y = sum_r( x[:, Kr] . W[:, Kr]^T ) with Kr this rank's slice of the augmented K axis.

    Two constructions: ``RowParallelARCLinear(QW, SFW, ...)`` slices a weight quantised as a whole (``shard_k``; the activation is
    quantised on the full row and sliced, hand-off A), ``RowParallelARCLinear.from_local(...)`` takes a shard that was quantised
    by itself with a shard-local reorder_index (hand-off B; ``forward_local`` quantises this rank's activation columns with the
    GLOBAL per-tensor scale and contracts them in one launch for decode-sized M).
    Partials: fp32 + one all-reduce for M <= 16, bf16 + two-shot (reduce-scatter + all-gather) above; the bias and the residual
    are added AFTER the reduction with the single-GPU epilogue's roundings (``finish_row_parallel``)."""

    def __init__(self, QW, SFW, scale_w, rank: int, world: int, bias=None, group=None, ops=None):
        self.rank, self.world, self.group, self.ops = rank, world, group, _ops(ops)
        self.K = QW.shape[1] * 2
        self.N = QW.shape[0]
        self.k0, self.k1 = k_slices(self.K, world)[rank]
        self.W, self.SFW = shard_k(QW, SFW, self.k0, self.k1)
        self.scale_w = scale_w
        self.bias = bias
        self.RW = self.RSF = None
        self.local_index = None
        self.KE_local = 0

    @classmethod
    def from_local(cls, QW_local, SFW_local, scale_w, local_index, KE_local, rank: int, world: int, bias=None, group=None, ops=None):
        self = cls.__new__(cls)
        self.rank, self.world, self.group, self.ops = rank, world, group, _ops(ops)
        self.K = QW_local.shape[1] * 2                         # the shard's own augmented K
        self.N = QW_local.shape[0]
        self.k0, self.k1 = 0, self.K
        self.W, self.SFW = QW_local, SFW_local
        self.scale_w, self.bias = scale_w, bias
        self.RW = self.RSF = None
        self.local_index, self.KE_local = local_index, int(KE_local)
        return self

    def repack_for_decode(self):
        if self.k1 > self.k0:
            self.RW, self.RSF = self.ops.repack_w(self.W, self.SFW)
        return self

    def shard_activation(self, qx, sfx):
        """The activation is quantised on the FULL row (its per-16 scales and residual channels do not
        depend on the split) and then sliced to this rank's K range."""
        return shard_k(qx, sfx, self.k0, self.k1)

    def _reduce_finish(self, part, residual, out_dtype):
        if self.world > 1:
            all_reduce_sum(part, group=self.group)             # RCCL over xGMI
        if out_dtype == torch.float32:
            if self.bias is not None:
                part = part + self.bias.float()
            return part if residual is None else part + residual.float()
        return finish_row_parallel(part, self.bias, residual)

    def _partial(self, a, sfa, scale_x, M):
        ops = self.ops
        pdtype = torch.float32 if M <= DECODE_M else torch.bfloat16
        if self.k1 == self.k0:                                 # more ranks than K atoms: this rank contributes nothing
            return torch.zeros((M, self.N), dtype=pdtype, device=a.device)
        scale, scale_host = _alpha(scale_x, self.scale_w)
        if self.RW is not None and ops.repacked_supported(M, self.N, self.k1 - self.k0):
            return ops.matmul_repacked(a, self.RW, sfa, self.RSF, scale, self.N, out_dtype=pdtype, scale_host=scale_host)
        return ops.matmul(a, self.W, sfa, self.SFW, scale, out_dtype=pdtype, scale_host=scale_host)

    def forward(self, qx, sfx, scale_x, out_dtype=torch.bfloat16, residual=None):
        """Hand-off A: (qx, sfx) is the activation quantised on the FULL row."""
        a, sfa = self.shard_activation(qx, sfx)
        return self._reduce_finish(self._partial(a, sfa, scale_x, qx.shape[0]), residual, out_dtype)

    def forward_local(self, x_local, word, out_dtype=torch.bfloat16, residual=None):
        """Hand-off B: ``x_local`` bf16 [M, K_local columns of this rank], ``word`` the GLOBAL abs-max word (``handoff_local_scale``).
        Decode-sized M: quantiser + GEMM in one launch (``dynamic_matmul_repacked`` with the word as its one abs-max slot)."""
        ops, M, KQ = self.ops, x_local.shape[0], x_local.shape[1]
        if M <= DECODE_M and self.RW is not None and ops.fused_supported(ops.SRC_DYNAMIC, M, self.N, KQ, self.KE_local):
            part, _ = ops.dynamic_matmul_repacked(x_local, self.local_index, self.KE_local, self.RW, self.RSF, float(self.scale_w), self.N,
                                                  absmax_slots=word, out_dtype=torch.float32)
        else:
            qa, sfa, sa = ops.reorder_quantize_x_dynamic(x_local, self.local_index, self.KE_local, absmax_slots=word)
            part = self._partial(qa, sfa, sa, M)
        return self._reduce_finish(part, residual, out_dtype)


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


def handoff_local_scale(y_local: torch.Tensor = None, group=None, word: torch.Tensor = None) -> torch.Tensor:
    """Hand-off B: the GLOBAL per-tensor abs-max word from the rank-local ones: one all-reduce(MAX) of 4 bytes.  Feed it to
    ``agemm.reorder_quantize_x_dynamic(y_local, local_reorder_index, KE_local, absmax_slots=word)``: the rank quantises only
    its own columns, with the scale the unsharded layer would use.  ``word``: the local word when the producing kernel already
    left it (``ColumnParallelARCLinear.forward_rmsnorm_silu``)."""
    import torch.distributed as dist
    if word is None:
        word = absmax_word(y_local)
    word = word.clone()
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


# ---- a tensor-parallel decoder layer (BASELINE config[4]: Llama-3-70B, TP = 8) ------------------------------------------------
class TPDecoderLayer:
    """One Llama-style decoder layer on ONE rank of a tensor-parallel group, decode step (one new token per sequence):

        RMSNorm+quantise -> q|k|v (column shard: this rank's heads) -> attention over the rank's KV heads
        -> [hand-off B: all-reduce(MAX) of 4 B] -> o_proj (row shard, local reorder_index) -> all-reduce(SUM) -> + residual
        RMSNorm+quantise -> gate|up (column shard, rows interleaved) -> SiLU*up
        -> [hand-off B: all-reduce(MAX) of 4 B] -> down_proj (row shard) -> all-reduce(SUM) -> + residual

    Per layer and rank on the wire: 2 x 4 B (MAX) + 2 x M x hidden x 4 B (SUM of fp32 partials, M <= 16).  The reference has no
    counterpart (model/parallel_utils.py:89-163 only places whole layers on devices); the single-GPU structure it mirrors is
    benchmarks/modeling_arc.py:279-310.  With ``repack_for_decode()`` every linear is ONE launch (quantiser as the GEMM's prologue).
    Attention is harness glue (torch SDPA over a dense bf16 cache of the rank's KV heads, GQA by grouping, no RoPE -- as e2e.py).
    ``ops`` is the operator module (``arcquant_amd.agemm``; the CPU tests pass an oracle-backed stand-in with the same functions)."""

    def __init__(self, qkv: "ColumnParallelARCLinear", o: "RowParallelARCLinear", gateup: "ColumnParallelARCLinear", down: "RowParallelARCLinear",
                 ln1, ln2, idx_h, KE: int, heads_local: int, kv_heads_local: int, head_dim: int, batch: int, max_len: int, eps: float = 1e-5,
                 group=None):
        self.qkv, self.o, self.gateup, self.down = qkv, o, gateup, down
        self.ln1, self.ln2, self.idx_h, self.KE, self.eps, self.group = ln1, ln2, idx_h, int(KE), float(eps), group
        self.hq, self.hk, self.hd = heads_local, kv_heads_local, head_dim
        dev = ln1.device
        self.kv = torch.zeros((2, batch, kv_heads_local, max_len, head_dim), dtype=torch.bfloat16, device=dev)
        self.trace = None                                      # a dict: the stage outputs of the last forward (tests)

    @staticmethod
    def shard_weights(dense: dict, rank: int, world: int, heads: int, kv_heads: int, head_dim: int):
        """This rank's bf16 shards of a dense layer {wq, wk, wv, wo, wg, wu, wd}: q|k|v rows of its heads, o columns of its
        heads, gate|up rows INTERLEAVED (g0, u0, g1, u1, ...) of its intermediate slice, down columns of that slice."""
        hq, hk = heads // world * head_dim, kv_heads // world * head_dim
        inter = dense["wg"].shape[0] // world
        wqkv = torch.cat([dense["wq"][rank * hq:(rank + 1) * hq], dense["wk"][rank * hk:(rank + 1) * hk], dense["wv"][rank * hk:(rank + 1) * hk]], dim=0)
        g, u = dense["wg"][rank * inter:(rank + 1) * inter], dense["wu"][rank * inter:(rank + 1) * inter]
        wgu = torch.stack([g, u], dim=1).reshape(2 * inter, -1)
        return dict(wqkv=wqkv.contiguous(), wo=dense["wo"][:, rank * hq:(rank + 1) * hq].contiguous(), wgu=wgu.contiguous(),
                    wd=dense["wd"][:, rank * inter:(rank + 1) * inter].contiguous())

    @classmethod
    def build(cls, shards: dict, ln1, ln2, idx_h, idx_o_local, idx_d_local, KE: int, KE_o: int, KE_d: int, rank: int, world: int, heads: int,
              kv_heads: int, head_dim: int, batch: int, max_len: int, eps: float = 1e-5, scales=None, group=None, ops=None, repack: bool = True):
        """Quantise this rank's shards (``shard_weights``) with ``ops.reorder_quantize_w`` and wire the four linears.  ``scales``:
        per-tensor weight scales {wqkv, wo, wgu, wd} (floats); default = the shard's own ``max(w) / 2688`` (qLinearLayer.py:25-28)."""
        ops = _ops(ops)
        lin = {}
        for name, idx, ke in (("wqkv", idx_h, KE), ("wo", idx_o_local, KE_o), ("wgu", idx_h, KE), ("wd", idx_d_local, KE_d)):
            w = shards[name]
            s = float(scales[name]) if scales is not None else float(torch.max(w).float() / (448.0 * 6.0))
            q, sf = ops.reorder_quantize_w((w.float() / s).to(torch.bfloat16).contiguous(), idx, ke)
            lin[name] = (q, sf, s)
        qkv = ColumnParallelARCLinear.from_local(*lin["wqkv"], rank, world, group=group, ops=ops)
        gateup = ColumnParallelARCLinear.from_local(*lin["wgu"], rank, world, group=group, ops=ops)
        o = RowParallelARCLinear.from_local(*lin["wo"], idx_o_local, KE_o, rank, world, group=group, ops=ops)
        down = RowParallelARCLinear.from_local(*lin["wd"], idx_d_local, KE_d, rank, world, group=group, ops=ops)
        if repack:
            for m in (qkv, gateup, o, down):
                m.repack_for_decode()
        return cls(qkv, o, gateup, down, ln1, ln2, idx_h, KE, heads // world, kv_heads // world, head_dim, batch, max_len, eps, group)

    def attention(self, qkv: torch.Tensor, pos: int) -> torch.Tensor:
        """Append this token's k / v of the rank's KV heads at `pos`, attend over [0, pos]: bf16 [batch, heads_local * head_dim]."""
        B, hq, hk, hd = qkv.shape[0], self.hq, self.hk, self.hd
        q = qkv[:, : hq * hd].reshape(B, hk, hq // hk, hd)                      # the query heads of one KV head form a "sequence"
        self.kv[0, :, :, pos] = qkv[:, hq * hd: (hq + hk) * hd].reshape(B, hk, hd)
        self.kv[1, :, :, pos] = qkv[:, (hq + hk) * hd:].reshape(B, hk, hd)
        att = torch.nn.functional.scaled_dot_product_attention(q, self.kv[0, :, :, : pos + 1], self.kv[1, :, :, : pos + 1])
        return att.reshape(B, hq * hd).contiguous()

    def forward(self, h: torch.Tensor, pos: int, trace: bool = False) -> torch.Tensor:
        """h: bf16 [batch, hidden], replicated; returns the layer's output, identical on every rank."""
        t = {} if trace else None
        qkv = self.qkv.forward_rmsnorm(h, self.ln1, self.eps, self.idx_h, self.KE)
        att = self.attention(qkv, pos)
        word_o = handoff_local_scale(att, group=self.group)
        h1 = self.o.forward_local(att, word_o, residual=h)
        act, word_l = self.gateup.forward_rmsnorm_silu(h1, self.ln2, self.eps, self.idx_h, self.KE)
        word_d = handoff_local_scale(group=self.group, word=word_l)
        h2 = self.down.forward_local(act, word_d, residual=h1)
        if trace:
            t.update(qkv=qkv, att=att, word_o=word_o, h1=h1, act=act, word_local=word_l, word_d=word_d, h2=h2)
            self.trace = t
        return h2

    @staticmethod
    def collective_bytes_per_layer(M: int, hidden: int, world: int) -> dict:
        """What one rank contributes per layer and decode step: message sizes (a ring moves 2 (p-1)/p of a SUM message per rank)."""
        return {"allreduce_max_word": 2 * 4, "allreduce_sum_fp32_partials": 2 * M * hidden * 4,
                "ring_bytes_on_the_wire_per_rank": int(2 * (2 * (world - 1) / world) * M * hidden * 4)}
