"""How fast can torch itself do one decode-attention call of the harness (bs=4, 28 heads x 128, 1040 cached tokens, bf16)?"""
import torch, torch.nn.functional as F, time
from torch.nn.attention import sdpa_kernel, SDPBackend
dev = "cuda:0"
B, H, T, D = 4, 28, 1040, 128
q = torch.randn(B, H, 1, D, device=dev, dtype=torch.bfloat16)
kc = torch.randn(B, H, T + 100, D, device=dev, dtype=torch.bfloat16)
vc = torch.randn(B, H, T + 100, D, device=dev, dtype=torch.bfloat16)
k, v = kc[:, :, :T], vc[:, :, :T]
def timeit(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5): g.replay()
    e0.record()
    for _ in range(n // 10): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (n // 10 * 10)
print("default sdpa", round(timeit(lambda: F.scaled_dot_product_attention(q, k, v)), 2), "us")
for name, be in (("flash", SDPBackend.FLASH_ATTENTION), ("efficient", SDPBackend.EFFICIENT_ATTENTION), ("math", SDPBackend.MATH)):
    try:
        with sdpa_kernel([be]):
            print(name, round(timeit(lambda: F.scaled_dot_product_attention(q, k, v)), 2), "us")
    except Exception as e:
        print(name, "failed:", str(e)[:100])
def bmm():
    sc = torch.matmul(q, k.transpose(2, 3)) * (D ** -0.5)
    return torch.matmul(torch.softmax(sc, dim=-1), v)
print("bmm+softmax (bf16)", round(timeit(bmm), 2), "us")
# queries of all heads as rows of ONE matmul per batch? (K differs per head: no)  -- 8 query rows padded
q8 = q.expand(B, H, 8, D).contiguous()
print("sdpa q_len 8 (padding)", round(timeit(lambda: F.scaled_dot_product_attention(q8, k, v)), 2), "us")
kt = kc.transpose(2, 3).contiguous()[:, :, :, :T]
print("bytes", 2 * B * H * T * D * 2 / 1e6, "MB -> at 5 TB/s", round(2 * B * H * T * D * 2 / 5e6, 1), "us")
