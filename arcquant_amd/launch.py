"""Start one process per GPU of this node (``bench.py --gpus N``, ``python -m arcquant_amd.e2e --tp N``) when no launcher did.

The parent that calls ``launch_ranks`` must not have touched the GPU (no ``torch.cuda`` call, no HIP call: on the MI355X pool an
``exec`` from a GPU-initialised process is refused); the ranks are plain child processes with the usual ``torch.distributed``
environment (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR=127.0.0.1, MASTER_PORT) and ``HSA_ENABLE_IPC_MODE_LEGACY=0`` (the host
driver only supports dmabuf IPC; RCCL needs it).  Rank 0's standard output is forwarded, the others' is dropped; the first rank
that fails ends the others (exact PIDs) and its exit code is returned."""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import tempfile
import time


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launched() -> bool:
    """True inside a rank process (started by ``launch_ranks`` or by ``torch.distributed.run``)."""
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def launch_ranks(n: int, cmd, env_extra=None, need_json: bool = True, timeout_s: float = None) -> int:
    """Run ``cmd`` (an argv list; usually ``[sys.executable, script] + sys.argv[1:]``) as ``n`` ranks.  Returns the exit code."""
    port = free_port()
    procs, out0 = [], None
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", ARCQ_RANKS_LAUNCHED="1")
        if env_extra:
            env.update(env_extra)
        if r == 0:
            out0 = tempfile.TemporaryFile(mode="w+")             # a file, not a pipe: nobody has to drain it while we poll
        procs.append(subprocess.Popen(list(cmd), env=env, stdout=out0 if r == 0 else subprocess.DEVNULL))
    rc, t0 = 0, time.monotonic()
    try:
        live = set(range(n))
        while live:
            for r in sorted(live):
                code = procs[r].poll()
                if code is None:
                    continue
                live.discard(r)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    sys.stderr.write(f"launch_ranks: rank {r} exited with code {code}; ending the other ranks\n")
                    for o in live:
                        procs[o].terminate()
            if timeout_s is not None and time.monotonic() - t0 > timeout_s and rc == 0:
                rc = 124
                sys.stderr.write(f"launch_ranks: no result after {timeout_s:.0f} s; ending the ranks\n")
                for o in live:
                    procs[o].terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    out0.seek(0)
    text = out0.read()
    out0.close()
    if need_json:         # the contract is ONE JSON line on standard output: whatever else rank 0 (or a library under it) printed goes to stderr
        for line in text.splitlines(keepends=True):
            (sys.stdout if line.startswith("{") else sys.stderr).write(line)
    else:
        sys.stdout.write(text)
    sys.stdout.flush()
    if rc == 0 and need_json and not any(line.startswith("{") for line in text.splitlines()):
        sys.stderr.write("launch_ranks: rank 0 printed no JSON line\n")
        rc = 1
    return rc
