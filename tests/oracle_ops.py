"""TEST INFRASTRUCTURE: the function surface ``arcquant_amd.tp`` uses of ``arcquant_amd.agemm``, answered by the CPU oracle on CPU
tensors -- so that the tensor-parallel host logic (sharding, hand-offs, collectives over gloo) runs in this container, where the
product GEMM (a GPU kernel) cannot.  Never imported by the package; the product path has no CPU fallback."""
import functools

import numpy as np
import torch

from oracle import oracle as O
from tests.util import bits, from_bits

SRC_RMSNORM, SRC_DYNAMIC = 1, 2
G32_KQ = (3584, 18944, 27648, 28672)          # bindings.cpp:141-160 (arcq_variant_for_kq)


@functools.lru_cache(maxsize=None)
def variant_for_kq(KQ):
    return O.G32 if KQ in G32_KQ else O.G16


def repacked_supported(M, N, K):
    return False


def fused_supported(kind, M, N, KQ, KE):
    return False


def _u8(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def reorder_quantize_w(W, reorder_index, KE, variant=None):
    q, sf = O.quantize_w(bits(W), reorder_index.numpy(), int(KE), variant_for_kq(W.shape[1]) if variant is None else variant, sf_fill=0)
    return _u8(q), _u8(sf)


def reorder_quantize_x(X, reorder_index, KE, variant=None):
    q, sf = O.quantize_x(bits(X), reorder_index.numpy(), int(KE), variant_for_kq(X.shape[1]) if variant is None else variant, sf_fill=0)
    return _u8(q), _u8(sf)


def rmsnorm_quantize_x(X, W, eps, reorder_index, KE, variant=None):
    q, sf = O.rmsnorm_quantize_x(bits(X), bits(W), float(eps), reorder_index.numpy(), int(KE),
                                 variant_for_kq(X.shape[1]) if variant is None else variant, sf_fill=0)
    return _u8(q), _u8(sf)


def dyn_scale_and_prescale(x, word):
    """scale = max|x| / 2688 and bf16(x / scale) with torch-on-GPU semantics (tests/test_gpu_parity._oracle_dyn_quant): the scale
    is amax * fl(1/2688), the division is by the scale rounded to bf16, quotient formed in fp32 and rounded once."""
    amax = torch.tensor([int(word)], dtype=torch.int32).to(torch.int16).view(torch.bfloat16).float()[0]
    scale = amax * torch.tensor(1.0 / 2688.0, dtype=torch.float32)
    xs = (x.float() / scale.to(torch.bfloat16).float()).to(torch.bfloat16)
    return scale, xs


def reorder_quantize_x_dynamic(X, reorder_index, KE, variant=None, absmax_slots=None):
    word = int(absmax_slots.max()) if absmax_slots is not None else int((X.contiguous().view(torch.int16).to(torch.int32) & 0x7FFF).max())
    scale, xs = dyn_scale_and_prescale(X, word)
    q, sf = reorder_quantize_x(xs.contiguous(), reorder_index, KE, variant)
    return q, sf, scale.reshape(())


def matmul(A, B, SFA, SFB, scale, *, bias=None, residual=None, out_dtype=torch.bfloat16, out=None, scale_host=1.0):
    alpha = np.float32(float(scale)) * np.float32(float(scale_host))
    db, de = O.gemm(A.numpy(), B.numpy(), SFA.numpy(), SFB.numpy(), alpha)
    if out_dtype == torch.float32:
        y = torch.from_numpy(de.astype(np.float32))
        if bias is not None:
            y = y + bias.float()
        if residual is not None:
            y = y + residual.float()
        return y
    y = from_bits(db)
    if bias is not None:
        y = y + bias
    if residual is not None:
        y = residual + y
    return y
