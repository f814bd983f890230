"""Shared test helpers (synthetic inputs follow kernels/main.py:13-19 of the reference)."""
import numpy as np
import torch


def bits(t: torch.Tensor) -> np.ndarray:
    """bf16/fp16 tensor -> uint16 bit patterns (numpy, on host)."""
    return t.detach().cpu().contiguous().view(torch.int16).numpy().view(np.uint16).copy()


def from_bits(b: np.ndarray, dtype=torch.bfloat16) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(b).view(np.int16)).view(dtype)


def outlier_activations(M, K, seed, dtype=torch.bfloat16):
    """Structured-outlier activations: uniform*3 with bands (x3+3, x8+8, x32+32) on the last channels."""
    g = torch.Generator().manual_seed(seed)
    ks, ko = max(16, K * 384 // 4096), max(16, K * 128 // 4096)
    signs = torch.randint(0, 2, (M, K), generator=g).to(dtype) * 2 - 1
    x = torch.rand(M, K, generator=g).to(dtype) * 3
    x[:, -ks:] = torch.rand(M, ks, generator=g).to(dtype) * 3 + 3
    x[:, -ko:] = torch.rand(M, ko, generator=g).to(dtype) * 8 + 8
    x[:, -16:] = torch.rand(M, 16, generator=g).to(dtype) * 32 + 32
    return x * signs


def prescale(x: torch.Tensor):
    """Per-tensor pre-scale of the callers (model/qLlamaLayer.py:73-77): x / (max|x| / 2688)."""
    s = x.abs().max().float() / (448.0 * 6.0)
    return (x / s).to(x.dtype), s


def random_perm(K, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randperm(K, generator=g).to(torch.int16)


def run_program(cmd, env, cwd, out_path, timeout=900):
    """Target of a fork-server process (tests/conftest.py FORKSERVER): run a program from a process that never touched the GPU and
    leave its exit code and output in a JSON file.  Lives here so that the child imports nothing but this module."""
    import json
    import subprocess
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, cwd=cwd, timeout=timeout)
    with open(out_path, "w") as f:
        json.dump({"rc": r.returncode, "stdout": r.stdout, "stderr": r.stderr[-4000:]}, f)
