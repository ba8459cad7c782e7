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


def test_two_ranks_share_the_gpu_over_gloo(gpu):
    """Rehearsal of the N > 1 launch path of bench.py on the one-GPU box: `torch.distributed.run --nproc-per-node 2
    bench.py --gpus 2` with both ranks on cuda:0 and the gradient buckets all-reduced by gloo (RCCL refuses two ranks
    on one device): episode sharding, the hooked last episode with bucket-ordered collectives on HIP tensors, barrier +
    MAX-over-ranks timing, rank 0's JSON line."""
    env = dict(os.environ, FPSG_DIST_BACKEND="gloo", FPSG_LOCAL_DEVICE="0", PYTHONPATH=ROOT)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), "bench.py", "--gpus", "2",
                        "--workload", "c3", "--steps", "2", "--warmup", "1", "--episodes-per-rank", "2"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                   # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["episodes_per_step_global"] == 4 and d["scaling"] == "weak"
    assert d["value"] > 0 and d["final_loss"] == d["final_loss"]
    assert "cpu_baseline" not in d and "configs" not in d    # N = 1 legs only
    assert len(d["allreduce"]["buckets"]) == 4


def test_bench_typed_with_gpus_2_launches_itself(gpu):
    """`python3 bench.py --gpus 2` as the driver types it for N = 1 -- no launcher, no torchrun variables: bench.py starts
    torch.distributed.run as a child process before it touches the GPU and relays rank 0's one line and the exit code."""
    env = dict(os.environ, FPSG_DIST_BACKEND="gloo", FPSG_LOCAL_DEVICE="0", PYTHONPATH=ROOT)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--workload", "c3", "--steps", "2", "--warmup", "1"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "launching" in r.stderr and "torch.distributed.run" in r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["distributed"]["world_size"] == 2 and d["distributed"]["ranks_that_reported"] == 2
    assert d["config"]["episodes_per_step_global"] == 2 and d["value"] > 0
