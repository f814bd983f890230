"""Synthetic-weight decode / prefill harness for the ARC-NVFP4 hot path.

Reproduces the CALLER side of the reference's latency benchmark -- benchmarks/modeling_arc.py (decoder layer
structure :279-310, MLP :109-111, RMSNorm+quantise :211-228, QLinearLayer :19-56) and
benchmarks/benchmark_e2e_arc.py (MODEL_CFGS :26-77, protocol :84-115) -- on top of ``arcquant_amd.agemm``:

    per layer:  rmsnorm_quantize_x -> q/k/v GEMMs -> attention -> reorder_quantize_x -> o GEMM -> +residual
                rmsnorm_quantize_x -> gate/up GEMMs -> silu*mul -> reorder_quantize_x -> down GEMM -> +residual

Differences from the reference files (which cannot run here: they need the `flashinfer` package and
transformers-4.44 internals): attention is torch SDPA over a dense bf16 KV cache instead of flashinfer paged KV;
weights are random-init and quantised once with ``agemm.reorder_quantize_w`` (the reference uses all-zero codes);
``select_num`` is 64 for every linear (the calibration artefact is not in the repo); the decode step is captured
in a HIP graph, so host launch overhead does not pace the GPU (every arcquant_amd launch is stream-ordered and
allocation-free apart from torch's caching allocator).
"""
from __future__ import annotations

import dataclasses
import time

import torch
import torch.nn.functional as F

from . import agemm


@dataclasses.dataclass
class ModelConfig:
    name: str
    num_layers: int
    num_heads: int
    hidden_size: int
    intermediate_size: int
    vocab_size: int = 32000          # LlamaConfig default used by benchmarks/modeling_arc.py:431
    select_num: int = 64
    eps: float = 1e-6
    attention_bias: bool = False     # benchmarks/benchmark_e2e_arc.py:21-22, 27-77: q/k/v/o (modeling_arc.py:19-56)
    mlp_bias: bool = False           # gate/up/down


# benchmarks/benchmark_e2e_arc.py:26-77
MODEL_CFGS = {
    "qwen2.5-7b": ModelConfig("qwen2.5-7b", 28, 28, 3584, 18944, attention_bias=True, mlp_bias=True),
    "llama-2-7b": ModelConfig("llama-2-7b", 32, 32, 4096, 11008),
    "llama-3.1-8b": ModelConfig("llama-3.1-8b", 32, 32, 4096, 14336),
    "qwen2.5-14b": ModelConfig("qwen2.5-14b", 48, 40, 5120, 13824, attention_bias=True, mlp_bias=True),
    "qwen2.5-32b": ModelConfig("qwen2.5-32b", 64, 40, 5120, 27648, attention_bias=True, mlp_bias=True),
}


class QLinear:
    """Quantised weight of one linear (random init), model/qLinearLayer.py:30-62 without the nn.Module shell.
    ``out_f`` may be the concatenation of several projections that share their input (q|k|v, gate|up): they are
    then quantised as ONE tensor (one per-tensor scale) and computed by one GEMM launch."""

    def __init__(self, in_f, out_f, select_num, device, gen, bias=False):
        w = (torch.randn(out_f, in_f, generator=gen, device=device, dtype=torch.float32) * 0.02).to(torch.bfloat16)
        # modeling_arc.py:37-40: QLinearLayer(bias=config.attention_bias | mlp_bias); added after the GEMM (`y = y + self.bias`)
        self.bias = (torch.randn(out_f, generator=gen, device=device, dtype=torch.float32) * 0.02).to(torch.bfloat16) if bias else None
        self.idx = torch.arange(in_f, dtype=torch.int16, device=device)
        scale = torch.max(w).float() / (448.0 * 6.0)
        self.W, self.SFW = agemm.reorder_quantize_w((w / scale).contiguous(), self.idx, select_num)
        self.scale = scale.reshape(1)
        self.scale_f = float(scale)          # one sync at load time; lets alpha = scale_f * (device activation scale)
        self.in_f, self.out_f, self.KE = in_f, out_f, select_num
        self.RW = self.RSF = None

    def repack(self):
        """Second copy of the weight in MFMA-operand-order tiles for the decode path (agemm.repack_w)."""
        self.RW, self.RSF = agemm.repack_w(self.W, self.SFW)

    def matmul(self, A, SFA, scale, ops=agemm, **kw):
        """GEMM against this weight (+ its bias, in the epilogue): the repacked kernel for decode-sized token counts where available.
        ``ops``: the module whose ``matmul_repacked`` is called (the ctypes mirror or the extension module)."""
        kw.setdefault("bias", self.bias)
        if self.RW is not None and agemm.repacked_supported(A.shape[0], self.out_f, self.in_f + self.KE):
            return ops.matmul_repacked(A, self.RW, SFA, self.RSF, scale, self.out_f, **kw)
        return agemm.matmul(A, self.W, SFA, self.SFW, scale, **kw)

    def bytes(self):
        return self.W.numel() + self.out_f * (self.in_f + self.KE) // 16


class DecoderModel:
    """fused=False: the reference's call structure (model/qLlamaLayer.py: separate q/k/v and gate/up GEMMs, torch
    abs/max/div before each activation quantise, torch residual adds).  fused=True: q|k|v and gate|up as one GEMM each,
    `reorder_quantize_x_dynamic` (1-2 launches instead of 5), SiLU*up in the gate|up GEMM epilogue (`matmul_silu_mul`),
    residual add in the GEMM epilogue, one strided K|V cache append."""

    def __init__(self, cfg: ModelConfig, batch: int, max_len: int, device, fused: bool = False, attention: str = "current"):
        """attention="current": what benchmarks/modeling_arc.py:169-198 times -- K/V are appended to the cache and each
        sequence attends (causally) over its CURRENT tokens only; attention="cache": attend over the whole KV cache."""
        self.cfg, self.device, self.batch, self.max_len, self.fused = cfg, device, batch, max_len, fused
        self.attention = attention
        # the down projection's quantiser as its GEMM's prologue: every CU then quantises the whole M x intermediate activation
        # itself, which only pays while that is small (measured: Qwen2.5-7B, 4 x 18944: slower than the separate launch)
        self.fused_down = batch * cfg.intermediate_size <= 4 * 8192
        # which decode linears run their activation quantiser as the GEMM's prologue (one launch instead of two); A-B'ed on
        # MI355X with tools/e2e_fuse_ab.py, the default is the fastest combination measured there
        import os
        self.fuse = set(filter(None, os.environ.get("ARCQ_E2E_FUSE", "qkv,o,gateup").split(",")))
        # the decode step's attention over the bf16 cache (harness glue, outside SURVEY 8): "stream" = the harness's own streaming
        # kernel (include/arcq_harness.h; it also appends k / v), "sdpa" / "bmm" = torch (45-48 us per layer at 1040 tokens)
        self.decode_attention = os.environ.get("ARCQ_E2E_DECODE_ATTENTION", "stream" if fused else "sdpa")
        self._attn_ws = None
        # the decode step's hot calls go through the extension module when it is built (same C-ABI entry points, ~half the host time per
        # call: an eager step is host-paced through ctypes); ARCQ_E2E_EXT=0 keeps the ctypes mirror
        self.fast = agemm
        if fused and os.environ.get("ARCQ_E2E_EXT", "1") == "1":
            try:
                from . import _build_ext
                self.fast = _build_ext.import_agemm_extension()
            except ImportError:
                pass
        g = torch.Generator(device=device).manual_seed(0)
        h, it, ke = cfg.hidden_size, cfg.intermediate_size, cfg.select_num
        ab, mb = cfg.attention_bias, cfg.mlp_bias
        self.layers = []
        for _ in range(cfg.num_layers):
            self.layers.append(dict(
                ln1=torch.ones(h, dtype=torch.bfloat16, device=device), ln2=torch.ones(h, dtype=torch.bfloat16, device=device),
                **(dict(qkv=QLinear(h, 3 * h, ke, device, g, ab), gateup=QLinear(h, 2 * it, ke, device, g, mb)) if fused else
                   dict(q=QLinear(h, h, ke, device, g, ab), k=QLinear(h, h, ke, device, g, ab), v=QLinear(h, h, ke, device, g, ab),
                        gate=QLinear(h, it, ke, device, g, mb), up=QLinear(h, it, ke, device, g, mb))),
                o=QLinear(h, h, ke, device, g, ab), down=QLinear(it, h, ke, device, g, mb),
                kv=torch.zeros(2, batch, cfg.num_heads, max_len, h // cfg.num_heads, dtype=torch.bfloat16, device=device)))
            self.layers[-1]["kc"], self.layers[-1]["vc"] = self.layers[-1]["kv"][0], self.layers[-1]["kv"][1]
        if fused:
            for L in self.layers:
                for name in ("qkv", "o", "gateup", "down"):
                    L[name].repack()
        self.idx_h = torch.arange(h, dtype=torch.int16, device=device)
        self.idx_i = torch.arange(it, dtype=torch.int16, device=device)
        self.inv_idx_i = torch.argsort(self.idx_i.long()).to(torch.int16)     # act_scatter_index of the gate|up epilogue
        agemm._check_scatter_index(self.inv_idx_i, it)                      # (the extension binding does not re-check the permutation)
        self.act_scatter = os.environ.get("ARCQ_E2E_ACT_SCATTER", "1") == "1"
        self.norm = torch.ones(h, dtype=torch.bfloat16, device=device)
        self.lm_head = (torch.randn(cfg.vocab_size, h, generator=g, device=device) * 0.02).to(torch.bfloat16)
        self.embed = (torch.randn(cfg.vocab_size, h, generator=g, device=device) * 0.02).to(torch.bfloat16)
        self.one = torch.ones(1, dtype=torch.float32, device=device)

    def weight_bytes(self):
        per_layer = sum(v.bytes() for v in self.layers[0].values() if isinstance(v, QLinear))
        return per_layer * self.cfg.num_layers + self.lm_head.numel() * 2

    @staticmethod
    def _quant_x(x, idx, ke):
        # model/qLlamaLayer.py:73-77 semantics, with the per-tensor scale kept on the device
        scale = agemm.absmax_scale(x)
        qx, sfx = agemm.reorder_quantize_x((x / scale).contiguous(), idx, ke)
        return qx, sfx, scale

    def forward(self, tokens: torch.Tensor, pos: int):
        """tokens [batch, q_len] int64; appends K/V at [pos, pos+q_len) and attends over [0, pos+q_len)."""
        cfg = self.cfg
        bsz, q_len = tokens.shape
        nh, hd = cfg.num_heads, cfg.hidden_size // cfg.num_heads
        hcur = F.embedding(tokens, self.embed).reshape(bsz * q_len, cfg.hidden_size)      # one gather launch (advanced indexing: ~10)
        h, it, ke = cfg.hidden_size, cfg.intermediate_size, cfg.select_num
        T = bsz * q_len
        for L in self.layers:
            # ---- attention block: RMSNorm + quantise, q|k|v projection
            if self.fused:
                Q = L["qkv"]
                if "qkv" in self.fuse and Q.RW is not None and agemm.fused_supported(agemm.SRC_RMSNORM, T, Q.out_f, h, ke):
                    # decode: ONE launch (the quantiser is the GEMM's prologue; bias in its epilogue)
                    qkv = self.fast.rmsnorm_matmul_repacked(hcur, L["ln1"], cfg.eps, self.idx_h, ke, Q.RW, Q.RSF, Q.scale, Q.out_f, bias=Q.bias)
                else:
                    A, SFA = agemm.rmsnorm_quantize_x(hcur, L["ln1"], cfg.eps, self.idx_h, ke)
                    qkv = Q.matmul(A, SFA, Q.scale)
                q, k, v = qkv[:, :h], qkv[:, h:2 * h], qkv[:, 2 * h:]
            else:
                A, SFA = agemm.rmsnorm_quantize_x(hcur, L["ln1"], cfg.eps, self.idx_h, ke)
                q, k, v = (self._ref_linear(L[n], A, SFA, L[n].scale) for n in ("q", "k", "v"))
            if self.fused and q_len == 1 and hd == 128 and self.decode_attention == "stream":
                att = self._attn_decode_stream(qkv, L, pos)          # K|V append + attention over [0, pos]: the harness's own kernel
            else:
                att = self._attention_torch(L, q, k, v, qkv if self.fused else None, pos, bsz, q_len)
            # ---- output projection (+ residual)
            if self.fused:
                O_ = L["o"]
                if "o" in self.fuse and O_.RW is not None and agemm.fused_supported(agemm.SRC_DYNAMIC, T, O_.out_f, h, ke):
                    hcur, _ = self.fast.dynamic_matmul_repacked(att, self.idx_h, ke, O_.RW, O_.RSF, O_.scale_f, O_.out_f, bias=O_.bias, residual=hcur)
                else:
                    qa, sfa, sa = agemm.reorder_quantize_x_dynamic(att, self.idx_h, ke)
                    hcur = O_.matmul(qa, sfa, sa, scale_host=O_.scale_f, residual=hcur)
            else:
                qa, sfa, sa = self._quant_x(att, self.idx_h, ke)
                hcur = hcur + self._ref_linear(L["o"], qa, sfa, sa * L["o"].scale)
            # ---- MLP
            if self.fused:
                Gt, D_ = L["gateup"], L["down"]         # one weight with gate and up rows interleaved (g0, u0, g1, u1, ...)
                slots = None
                if "gateup" in self.fuse and Gt.RW is not None and agemm.fused_supported(agemm.SRC_RMSNORM, T, Gt.out_f, h, ke):
                    # decode: RMSNorm + quantise + gate|up GEMM + bias + SiLU*up + abs-max words in ONE launch
                    # "scatter": its epilogue stores the activation in the DOWN projection's channel order, the quantiser below then
                    # reads contiguous groups (reorder_index=None) instead of staging + gathering each row
                    scatter = self.inv_idx_i if self.act_scatter and not (("down" in self.fuse or self.fused_down)) else None
                    act, slots = self.fast.rmsnorm_matmul_repacked_silu(hcur, L["ln2"], cfg.eps, self.idx_h, ke, Gt.RW, Gt.RSF, Gt.scale, Gt.out_f, bias=Gt.bias,
                                                                        act_scatter_index=scatter)
                    if scatter is not None:
                        qa, sfa, sa = self.fast.reorder_quantize_x_dynamic(act, None, ke, absmax_slots=slots)
                        hcur = D_.matmul(qa, sfa, sa, scale_host=D_.scale_f, residual=hcur, ops=self.fast)
                        continue
                else:
                    A, SFA = agemm.rmsnorm_quantize_x(hcur, L["ln2"], cfg.eps, self.idx_h, ke)
                    if T > 16:      # prefill: act_fn(gate) * up and its abs-max in the tile GEMM's epilogue
                        act, slots = agemm.matmul_silu_mul(A, Gt.W, SFA, Gt.SFW, Gt.scale, bias=Gt.bias)
                    elif Gt.RW is not None and Gt.bias is None and agemm.repacked_supported(T, Gt.out_f, h + ke):
                        # r1 path: the repacked GEMM leaves max |silu(g) * u| per row block, the quantiser applies SiLU*up itself
                        gu, slots = agemm.matmul_repacked_silu_absmax(A, Gt.RW, SFA, Gt.RSF, Gt.scale, Gt.out_f)
                        act = None
                    else:
                        gu = Gt.matmul(A, SFA, Gt.scale)
                        act = (F.silu(gu[:, 0::2]) * gu[:, 1::2]).contiguous()
                if act is None:
                    qa, sfa, sa = agemm.silu_mul_quantize_x_dynamic(gu, self.idx_i, ke, layout=agemm.GU_PAIRS, absmax_slots=slots)
                    hcur = D_.matmul(qa, sfa, sa, scale_host=D_.scale_f, residual=hcur)
                elif ("down" in self.fuse or self.fused_down) and D_.RW is not None and agemm.fused_supported(agemm.SRC_DYNAMIC, T, D_.out_f, it, ke):
                    hcur, _ = agemm.dynamic_matmul_repacked(act, self.idx_i, ke, D_.RW, D_.RSF, D_.scale_f, D_.out_f, absmax_slots=slots,
                                                            bias=D_.bias, residual=hcur)
                else:
                    qa, sfa, sa = agemm.reorder_quantize_x_dynamic(act, self.idx_i, ke, absmax_slots=slots)
                    hcur = D_.matmul(qa, sfa, sa, scale_host=D_.scale_f, residual=hcur)
            else:
                A, SFA = agemm.rmsnorm_quantize_x(hcur, L["ln2"], cfg.eps, self.idx_h, ke)
                gate = self._ref_linear(L["gate"], A, SFA, L["gate"].scale)
                up = self._ref_linear(L["up"], A, SFA, L["up"].scale)
                act = F.silu(gate) * up
                qa, sfa, sa = self._quant_x(act, self.idx_i, ke)
                hcur = hcur + self._ref_linear(L["down"], qa, sfa, sa * L["down"].scale)
        last = hcur.view(bsz, q_len, -1)[:, -1]                 # [bsz, hidden], row stride q_len * hidden
        if self.fused:
            # the final norm as one launch (include/arcq_harness.h; torch's F.rms_norm is ~15 small kernels on this stack)
            from . import _lib
            hn = torch.empty((bsz, cfg.hidden_size), dtype=torch.bfloat16, device=hcur.device)
            with torch.cuda.device(hcur.device):
                st = _lib.lib().arcq_harness_rmsnorm(last.data_ptr(), last.stride(0), self.norm.data_ptr(), hn.data_ptr(), bsz, cfg.hidden_size,
                                                     float(cfg.eps), torch.cuda.current_stream(hcur.device).cuda_stream)
            _lib.check(st, "harness rmsnorm")
        else:
            hn = F.rms_norm(last, (cfg.hidden_size,), self.norm, cfg.eps)
        return hn @ self.lm_head.t()

    def _attention_torch(self, L, q, k, v, qkv, pos, bsz, q_len):
        """KV append + attention with torch ops (prefill always; decode when the streaming kernel is switched off)."""
        cfg = self.cfg
        nh, hd, h = cfg.num_heads, cfg.hidden_size // cfg.num_heads, cfg.hidden_size
        q = q.reshape(bsz, q_len, nh, hd).transpose(1, 2)
        if qkv is not None:     # k|v are adjacent columns of the fused projection: ONE strided copy appends both to the cache
            L["kv"][:, :, :, pos:pos + q_len] = qkv[:, h:].reshape(bsz, q_len, 2, nh, hd).permute(2, 0, 3, 1, 4)
        else:
            L["kc"][:, :, pos:pos + q_len] = k.reshape(bsz, q_len, nh, hd).transpose(1, 2)
            L["vc"][:, :, pos:pos + q_len] = v.reshape(bsz, q_len, nh, hd).transpose(1, 2)
        if self.attention == "cache" and q_len == 1 and self.decode_attention == "bmm":
            kc, vc = L["kc"][:, :, :pos + 1], L["vc"][:, :, :pos + 1]
            sc = torch.matmul(q, kc.transpose(2, 3)) * (hd ** -0.5)
            att = torch.matmul(torch.softmax(sc.float(), dim=-1).to(q.dtype), vc)
        elif self.attention == "cache":
            att = F.scaled_dot_product_attention(q, L["kc"][:, :, :pos + q_len], L["vc"][:, :, :pos + q_len],
                                                 is_causal=(q_len > 1 and pos == 0))
        else:                   # benchmarks/modeling_arc.py:169-198: each sequence attends over the tokens of THIS call only
            att = F.scaled_dot_product_attention(q, k.reshape(bsz, q_len, nh, hd).transpose(1, 2),
                                                 v.reshape(bsz, q_len, nh, hd).transpose(1, 2), is_causal=q_len > 1)
        return att.transpose(1, 2).reshape(bsz * q_len, h)

    def _attn_decode_stream(self, qkv, L, pos):
        """One decode step of attention over the dense bf16 cache with the harness kernel (include/arcq_harness.h): appends this
        token's k / v at `pos` and attends over [0, pos] (attention="cache") or over the current token only ("current", what
        benchmarks/modeling_arc.py:169-198 times); fp32 math, bf16 out [batch, hidden]."""
        from . import _lib
        lib = _lib.lib()
        bsz, nh = qkv.shape[0], self.cfg.num_heads
        tmax = L["kc"].shape[2]
        if self._attn_ws is None:
            self._attn_ws = torch.empty(int(lib.arcq_harness_attn_workspace_bytes(bsz, nh, tmax)) // 4, dtype=torch.float32, device=qkv.device)
        out = torch.empty((bsz, self.cfg.hidden_size), dtype=torch.bfloat16, device=qkv.device)
        with torch.cuda.device(qkv.device):
            st = lib.arcq_harness_attn_decode_window(qkv.data_ptr(), L["kc"].data_ptr(), L["vc"].data_ptr(), out.data_ptr(), self._attn_ws.data_ptr(),
                                                     bsz, nh, tmax, int(pos), 0 if self.attention == "cache" else int(pos),
                                                     torch.cuda.current_stream(qkv.device).cuda_stream)
        _lib.check(st, "harness attn_decode")
        return out

    @staticmethod
    def _ref_linear(lin, A, SFA, scale):
        """The reference's QLinearLayer.forward (benchmarks/modeling_arc.py:47-56): GEMM, then `y = y + self.bias` as its own op."""
        y = agemm.matmul(A, lin.W, SFA, lin.SFW, scale)
        return y if lin.bias is None else y + lin.bias


def bench_decode(name="qwen2.5-7b", batch=4, prefill=1024, steps=16, device="cuda:0", repeats=3, layers=None, fused=False,
                 attention="current"):
    """Decode tok/s with the decode step replayed from a HIP graph (attention window fixed at prefill+steps)."""
    cfg = dataclasses.replace(MODEL_CFGS[name])
    if layers:
        cfg.num_layers = layers
    device = torch.device(device)
    with torch.no_grad():
        model = DecoderModel(cfg, batch, prefill + steps + 1, device, fused=fused, attention=attention)
        tok = torch.randint(100, 200, (batch, prefill), device=device)
        t0 = time.perf_counter()
        model.forward(tok, 0)
        torch.cuda.synchronize()
        t_prefill_first = time.perf_counter() - t0
        t0 = time.perf_counter()
        model.forward(tok, 0)
        torch.cuda.synchronize()
        t_prefill = time.perf_counter() - t0
        nxt = torch.randint(100, 200, (batch, 1), device=device)
        pos = prefill + steps          # fixed attention window: the graph is shape-static
        for _ in range(2):
            model.forward(nxt, pos)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            model.forward(nxt, pos)
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                out = model.forward(nxt, pos)
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        best = None
        for _ in range(repeats):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(steps):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / steps
            best = ms if best is None else min(best, ms)
        # eager (no graph) for comparison: host-paced
        t0 = time.perf_counter()
        for _ in range(4):
            model.forward(nxt, pos)
        torch.cuda.synchronize()
        eager_ms = (time.perf_counter() - t0) / 4 * 1e3
        wb = model.weight_bytes()
        kv = 2 * cfg.num_layers * batch * cfg.hidden_size * pos * 2 if attention == "cache" else 0
        assert out.shape == (batch, cfg.vocab_size)
    return {"model": name, "fused": fused, "attention": attention, "layers": cfg.num_layers, "batch": batch, "prefill": prefill, "attn_window": pos,
            "decode_ms_per_step_graph": round(best, 4), "decode_tok_per_s": round(batch / best * 1e3, 1),
            "decode_ms_per_step_eager": round(eager_ms, 3), "prefill_ms": round(t_prefill * 1e3, 2),
            "prefill_tok_per_s": round(batch * prefill / t_prefill, 0), "prefill_first_call_ms": round(t_prefill_first * 1e3, 1),
            "weight_bytes": wb, "kv_bytes_read_per_step": kv,
            "hbm_floor_ms_at_8TBps": round((wb + kv) / 8e12 * 1e3, 4)}


def bench_protocol(name="qwen2.5-7b", batch=4, prefill=1024, decode_steps=128, device="cuda:0", repeats=10, warmup=2, steps=4,
                   fused=True, attention="cache", graph=True, layers=None):
    """The reference's latency protocol (benchmarks/benchmark_e2e_arc.py): three timed modules -- prefill (:133-140), decode
    for `decode_steps` steps over a GROWING cache (:142-155) and prefill + decode (:157-166) -- each run `warmup` times
    untimed and `steps` times timed between two device synchronisations, repeated `repeats` times (:81-115); reported as
    mean +- 1.96 sigma of the per-call milliseconds plus the peak device memory (:208-216).  graph=True replays the whole
    multi-step decode (every step at its own cache length) from ONE HIP graph; graph=False issues it eagerly like the
    reference does."""
    cfg = dataclasses.replace(MODEL_CFGS[name])
    if layers:
        cfg.num_layers = layers
    device = torch.device(device)

    def module_benchmark(fn):
        times, peaks = [], []
        for _ in range(repeats):
            for _ in range(warmup):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            torch.cuda.reset_peak_memory_stats()
            for _ in range(steps):
                fn()
            torch.cuda.synchronize()
            peaks.append(torch.cuda.max_memory_allocated())
            times.append((time.perf_counter() - t0) * 1e3 / steps)
        t = torch.tensor(times, dtype=torch.float64)
        return round(float(t.mean()), 3), round(1.96 * float(t.std(unbiased=False)), 3), max(peaks)

    with torch.no_grad():
        model = DecoderModel(cfg, batch, prefill + decode_steps, device, fused=fused, attention=attention)
        tok = torch.randint(100, 200, (batch, prefill), device=device)
        nxt = torch.full((batch, 1), 100, device=device, dtype=torch.int64)          # benchmark_e2e_arc.py:150

        def run_prefill():
            model.forward(tok, 0)

        def decode_eager():
            for i in range(decode_steps):
                model.forward(nxt, prefill + i)

        run_prefill()
        decode_eager()
        torch.cuda.synchronize()
        run_decode = decode_eager
        if graph:
            g = torch.cuda.CUDAGraph()
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                with torch.cuda.graph(g, stream=st):
                    decode_eager()
            torch.cuda.synchronize()
            run_decode = g.replay

        def run_e2e():
            run_prefill()
            run_decode()

        p_ms, p_ci, p_mem = module_benchmark(run_prefill)
        d_ms, d_ci, d_mem = module_benchmark(run_decode)
        e_ms, e_ci, e_mem = module_benchmark(run_e2e)
    return {"protocol": "benchmark_e2e_arc.py: %d warm-up + %d timed calls x %d repeats, mean +- 1.96 sigma" % (warmup, steps, repeats),
            "model": name, "layers": cfg.num_layers, "batch": batch, "prefill": prefill, "decode_steps": decode_steps, "fused": fused,
            "attention": attention, "decode_from_hip_graph": bool(graph),
            "prefill_ms": [p_ms, p_ci], "decode_ms": [d_ms, d_ci], "e2e_ms": [e_ms, e_ci],
            "prefill_tok_per_s": round(batch * prefill / p_ms * 1e3, 0), "decode_tok_per_s": round(batch * decode_steps / d_ms * 1e3, 1),
            "peak_memory_gb": round(max(p_mem, d_mem, e_mem) / 2 ** 30, 3)}


# ---- BASELINE config[4]: one Llama-3-70B-shape decoder layer per rank of a tensor-parallel group -------------------------------------
LLAMA3_70B = dict(hidden=8192, heads=64, kv_heads=8, head_dim=128, inter=28672, layers=80)


def build_tp_layer(rank, world, device, batch, max_len, cfg=None, KE=64, seed=0, ops=None, repack=True, group=None):
    """This rank's shard of ONE decoder layer with random weights (tp.TPDecoderLayer; the shards are drawn directly, seed + rank,
    each with its own per-tensor weight scale), identity reorder indices, ``select_num`` = KE for every linear (per shard for the
    row-parallel ones: hand-off B)."""
    from . import tp
    cfg = dict(LLAMA3_70B if cfg is None else cfg)
    h, hd = cfg["hidden"], cfg["head_dim"]
    hq, hk, it = cfg["heads"] // world * hd, cfg["kv_heads"] // world * hd, cfg["inter"] // world
    g = torch.Generator(device=device).manual_seed(seed * 1000 + rank)

    def rnd(n, k):
        return (torch.randn(n, k, generator=g, device=device, dtype=torch.float32) * 0.02).to(torch.bfloat16)

    shards = dict(wqkv=rnd(hq + 2 * hk, h), wo=rnd(h, hq), wgu=rnd(2 * it, h), wd=rnd(h, it))
    ones = torch.ones(h, dtype=torch.bfloat16, device=device)
    idx_h = torch.arange(h, dtype=torch.int16, device=device)
    idx_o = torch.arange(hq, dtype=torch.int16, device=device)
    idx_d = torch.arange(it, dtype=torch.int16, device=device)
    return tp.TPDecoderLayer.build(shards, ones, ones.clone(), idx_h, idx_o, idx_d, KE, KE, KE, rank, world, cfg["heads"], cfg["kv_heads"], hd,
                                   batch, max_len, eps=1e-5, group=group, ops=ops, repack=repack)


def bench_tp_layer(rank, world, device, batch=4, pos=1040, steps=50, warmup=10, cfg=None):
    """Decode step of ONE tensor-parallel decoder layer (Llama-3-70B shape by default): per-layer time with its four collectives
    (2 x all-reduce(MAX) of 4 B, 2 x all-reduce(SUM) of the fp32 [batch, hidden] partial), the same step with the collectives
    skipped (what the rank's four launches + attention cost alone), max over ranks.  Launches are issued eagerly."""
    import torch.distributed as dist
    from . import tp
    cfg = dict(LLAMA3_70B if cfg is None else cfg)
    with torch.no_grad():
        layer = build_tp_layer(rank, world, device, batch, pos + 8, cfg)
        h = (torch.randn(batch, cfg["hidden"], device=device, generator=torch.Generator(device=device).manual_seed(7)) * 0.5).to(torch.bfloat16)

        def timed(fn):
            for _ in range(warmup):
                fn()
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(steps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            t = torch.tensor([e0.elapsed_time(e1) * 1e3 / steps], dtype=torch.float64, device=device)
            if world > 1:
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())

        with_coll = timed(lambda: layer.forward(h, pos))
        out = layer.forward(h, pos)
        # the same launches without the exchange: the linears think they are alone (world = 1); results are then partial sums
        for m in (layer.o, layer.down):
            m.world = 1
        grp, layer.group = layer.group, None
        saved = tp.handoff_local_scale
        tp.handoff_local_scale = lambda y_local=None, group=None, word=None: (tp.absmax_word(y_local) if word is None else word)
        try:
            no_coll = timed(lambda: layer.forward(h, pos))
        finally:
            tp.handoff_local_scale = saved
            layer.group = grp
            for m in (layer.o, layer.down):
                m.world = world
        same = True
        if world > 1:                                          # the replicated hidden state must be bit-identical on every rank
            ref = out.clone()
            dist.broadcast(ref, src=0)
            flag = torch.tensor([int(torch.equal(ref, out))], device=device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            same = bool(flag.item())
    wbytes = sum(m.W.numel() * 9 // 8 for m in (layer.qkv, layer.o, layer.gateup, layer.down))
    return {"model": "llama-3-70b-shape decoder layer", "tp": world, "batch": batch, "attn_window": pos, "launches": "eager",
            "layer_us_with_collectives": round(with_coll, 1), "layer_us_without_collectives": round(no_coll, 1),
            "weight_bytes_per_rank": int(wbytes), "collective_bytes": tp.TPDecoderLayer.collective_bytes_per_layer(batch, cfg["hidden"], world),
            "output_identical_on_all_ranks": same,
            "decode_tok_per_s_80_layers_est": round(batch / (with_coll * cfg["layers"]) * 1e6, 1)}


def _tp_main(argv):
    """python -m arcquant_amd.e2e --tp N [--steps K]: one rank per GPU (self-launched, or under torch.distributed.run).
    Rehearsal on a one-GPU box: ARCQ_BENCH_ONE_DEVICE=1 ARCQ_BENCH_BACKEND=gloo."""
    import json
    import os
    import sys
    from . import launch
    n = int(argv[argv.index("--tp") + 1])
    steps = int(argv[argv.index("--steps") + 1]) if "--steps" in argv else 50
    if n > 1 and not launch.launched():
        sys.exit(launch.launch_ranks(n, [sys.executable, "-m", "arcquant_amd.e2e"] + list(argv)))
    world, rank, local_rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    one_device = os.environ.get("ARCQ_BENCH_ONE_DEVICE") == "1"
    backend = os.environ.get("ARCQ_BENCH_BACKEND", "nccl")
    dev_index = 0 if one_device else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    cfg = None
    if "--small" in argv:                                      # a quick functional pass (tests): 1/8 of the 70B layer in every dimension
        cfg = dict(hidden=2048, heads=16, kv_heads=2 * max(1, world), head_dim=128, inter=7168 // 8 * 8, layers=80)
    res = bench_tp_layer(rank, world, device, steps=steps, cfg=cfg)
    if rank == 0:
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    import json
    import sys
    if "--tp" in sys.argv:
        _tp_main(sys.argv[1:])
        sys.exit(0)
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    name = args[0] if args else "qwen2.5-7b"
    if "--protocol" in sys.argv:      # the reference's own benchmark protocol (growing cache, mean +- 1.96 sigma)
        for graph in (True, False):
            print(json.dumps(bench_protocol(name, graph=graph)), flush=True)
            torch.cuda.empty_cache()
    else:
        for fused, att in ((False, "current"), (True, "current"), (True, "cache")):
            print(json.dumps(bench_decode(name, fused=fused, attention=att)))
            torch.cuda.empty_cache()
