"""`python bench.py --gpus 2` and `python -m arcquant_amd.e2e --tp 2` exactly as a user / the driver starts them -- no launcher, no
WORLD_SIZE -- on this ONE-GPU box: ARCQ_BENCH_ONE_DEVICE=1 puts both ranks on cuda:0, ARCQ_BENCH_BACKEND=gloo replaces RCCL (which
needs one device per rank).  Everything else is the multi-GPU path: the GPU-free parent starts the ranks, each rank starts its
strong-scaling helper before it touches the GPU, the ranks rendezvous on 127.0.0.1, rank 0 prints ONE JSON line, rc 0.

The programs are started through the fork server of tests/conftest.py (this pytest process has initialised the GPU and must not
exec)."""
import json
import os
import sys

import pytest

from tests import conftest
from tests.util import run_program

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _start(cmd, tmp_path):
    assert conftest.FORKSERVER is not None, "the fork server is started by tests/conftest.py on a GPU box"
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(ARCQ_BENCH_ONE_DEVICE="1", ARCQ_BENCH_BACKEND="gloo")
    out = str(tmp_path / "result.json")
    p = conftest.FORKSERVER.Process(target=run_program, args=(cmd, env, ROOT, out))
    p.start()
    p.join(1000)
    assert p.exitcode == 0, f"runner exit code {p.exitcode}"
    with open(out) as f:
        return json.load(f)


def test_bench_gpus_2_starts_its_own_ranks_and_prints_one_json_line(tmp_path):
    r = _start([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--prewarm-ms", "0", "--no-cpu"], tmp_path)
    assert r["rc"] == 0, r["stderr"]
    lines = r["stdout"].splitlines()                          # exactly ONE line, the JSON: library chatter (gloo prints to fd 1) goes to stderr
    assert len(lines) == 1 and lines[0].startswith("{"), r["stdout"]
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["scaling"] == "weak" and res["steps"] == 3 and res["value"] > 0
    assert res["roofline"]["bound"] == "mfma" and res["roofline"]["achieved"] > 0
    ss = res["extra"]["strong_scaling"]                     # run in the helper processes; present whatever happened in there
    assert "gemm_4096" in ss or "error" in ss, ss
    if "gemm_4096" in ss and "error" not in ss["gemm_4096"]:
        assert ss["gemm_4096"]["column_parallel_allgather"]["us"] > 0 and ss["gemm_4096"]["row_parallel_bf16_allreduce"]["us"] > 0


def test_e2e_tp_2_runs_one_decoder_layer_per_rank_with_its_collectives(tmp_path):
    r = _start([sys.executable, "-m", "arcquant_amd.e2e", "--tp", "2", "--small", "--steps", "5"], tmp_path)
    assert r["rc"] == 0, r["stderr"]
    res = json.loads([l for l in r["stdout"].splitlines() if l.startswith("{")][0])
    assert res["tp"] == 2 and res["output_identical_on_all_ranks"] is True
    assert res["layer_us_with_collectives"] > 0 and res["layer_us_without_collectives"] > 0
