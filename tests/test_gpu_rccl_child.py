"""The RCCL path of bench.py / dicp_amd.dist on the GPU box (`-m gpu`): a FRESH child process under torch.distributed.run with one rank and
DICP_BENCH_FORCE_DIST=1 -- RCCL is initialised, the barriers, the max-over-ranks reduction and the pose all-gather run on the real backend, and the
line says so.  (The child initialises the GPU itself; this process only starts it and reads its line.  The N > 1 launcher path is covered on gloo ranks
by tests/test_bench_launcher.py; a multi-GPU RCCL run is the driver's to make.)"""
import json
import math
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_bench_under_torch_distributed_run_uses_rccl():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, DICP_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2", "--batch", "32", "--points", "4096", "--no-cpu-baseline", "--no-extra-legs"]
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert lines, res.stdout[-2000:]
    line = json.loads(lines[-1])
    assert line["ranks_seen"] == 1 and line["n_gpus"] == 1
    assert isinstance(line["rccl_version"], str) and line["rccl_version"].count(".") >= 1, line["rccl_version"]
    assert line["pose_allgather_ms"] is not None and math.isfinite(line["pose_allgather_ms"]) and line["pose_allgather_ms"] > 0.0
    assert line["finite"] and line["value"] > 0
