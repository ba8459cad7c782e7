"""RCCL path on the one GPU there is: a single-rank process group (FPSG_FORCE_DIST=1) makes
bench.py initialise NCCL(=RCCL), arm the gradient buckets, launch the bucketed all-reduces
from the autograd hooks and mix hipGraph replays with the eager last episode."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("extra", [["--no-graph"], ["--graph", "--episodes-per-rank", "3"]])
def test_single_rank_rccl_step(gpu, extra):
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(_free_port()), FPSG_FORCE_DIST="1", PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "bench.py", "--workload", "c2", "--steps", "3", "--warmup", "4",
                        "--no-cpu-baseline", "--no-extra"] + extra, cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["final_loss"] == d["final_loss"]
    assert len(d["allreduce"]["buckets"]) >= 1 and all(b["ms"] > 0 for b in d["allreduce"]["buckets"])
