"""arcquant_amd/launch.py: `bench.py --gpus N` / `python -m arcquant_amd.e2e --tp N` start their own ranks when no launcher did
(the driver runs `python3 bench.py --gpus N ...`).  CPU tests of the launcher logic with stub rank scripts; the real thing runs in
tests/test_bench_launch_gpu.py on the GPU box."""
import json
import os
import subprocess
import sys
import textwrap

from arcquant_amd import launch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _script(tmp_path, body):
    p = tmp_path / "rank.py"
    p.write_text(textwrap.dedent(body))
    return [sys.executable, str(p)]


def test_ranks_get_the_distributed_environment_and_rank0_line_is_forwarded(tmp_path, capfd):
    cmd = _script(tmp_path, """
        import json, os, sys
        import torch.distributed as dist
        dist.init_process_group("gloo")                      # env:// rendezvous on MASTER_ADDR / MASTER_PORT
        import torch
        t = torch.tensor([float(os.environ["RANK"]) + 1.0])
        dist.all_reduce(t)
        print(json.dumps({"rank": int(os.environ["RANK"]), "world": dist.get_world_size(), "sum": float(t), "addr": os.environ["MASTER_ADDR"],
                          "ipc": os.environ["HSA_ENABLE_IPC_MODE_LEGACY"], "local": os.environ["LOCAL_RANK"]}))
        dist.destroy_process_group()
    """)
    rc = launch.launch_ranks(3, cmd)
    out = capfd.readouterr().out
    assert rc == 0
    lines = [json.loads(l) for l in out.splitlines() if l.startswith("{")]
    assert lines == [{"rank": 0, "world": 3, "sum": 6.0, "addr": "127.0.0.1", "ipc": "0", "local": "0"}]      # ONE line: rank 0's


def test_a_failing_rank_ends_the_others_and_the_exit_code_is_nonzero(tmp_path, capfd):
    cmd = _script(tmp_path, """
        import os, sys, time
        if os.environ["RANK"] == "1":
            sys.exit(7)
        time.sleep(60)                                       # would hang the launcher if it did not end us
        print("{}")
    """)
    import time
    t0 = time.monotonic()
    rc = launch.launch_ranks(2, cmd)
    assert rc == 7 and time.monotonic() - t0 < 30
    assert "rank 1 exited with code 7" in capfd.readouterr().err


def test_a_silent_rank0_is_an_error(tmp_path):
    assert launch.launch_ranks(2, _script(tmp_path, "pass\n")) == 1
    assert launch.launch_ranks(2, _script(tmp_path, "pass\n"), need_json=False) == 0


def test_bench_launches_its_own_ranks_and_fails_loudly_without_a_gpu():
    """No GPU in this container: every rank dies at torch.cuda.set_device -> the launcher must return non-zero (never a silent
    CPU fallback), and it must have been the launcher, not the old `sys.exit("launch with torch.distributed.run")`."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-extra", "--no-cpu"],
                       env=env, capture_output=True, text=True, timeout=300)
    import torch
    if torch.cuda.device_count() == 0:
        assert r.returncode != 0
        assert "launch_ranks: rank" in r.stderr and "torch.distributed.run" not in r.stderr
    else:                                                     # on a GPU box this is simply the 2-rank run on devices 0 and 1 (or a failure on one GPU)
        assert r.returncode == 0 or "launch_ranks: rank" in r.stderr


def test_launched_detects_a_rank_environment(monkeypatch):
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("RANK", raising=False)
    assert not launch.launched()
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "1")
    assert launch.launched()
