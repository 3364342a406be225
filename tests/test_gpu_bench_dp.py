"""-m gpu: the N-rank control flow of bench.py on a one-GPU box -- two ranks share GPU 0 and exchange through gloo
(STGCN_BENCH_BACKEND=gloo STGCN_BENCH_SHARE_GPU=1).  The throughput of such a run means nothing; what is exercised is everything the
driver's 2/4/8-GPU runs go through that a single-rank run does not: rendezvous, per-rank device windows of the resident series, the
two-graph step around the all-reduce with the step counters folded onto the pack launch, max-over-ranks timing, rank-0-only JSON."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_on_one_gpu():
    env = dict(os.environ, STGCN_BENCH_BACKEND="gloo", STGCN_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29613", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "3",
           "--no-cpu-baseline", "--no-gpu-baseline", "--no-profile"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]          # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 64 and out["config"]["parallelism"] == "dp2"
    assert out["value"] > 0 and out["steps"] == 10 and out["scaling"] == "weak"
    assert out["config"]["launch"] == "hipGraph replay", out["config"]
    ar = out["config"]["allreduce"]
    assert ar["bytes"] > 900_000 and 0 < ar["exposed_fraction_of_step"] < 1
    # gloo collectives are host calls: the capture probe must say so and the step must stay in the two-graph form
    assert ar["captured_in_graph"] is False and "gloo" in ar["capture_probe"], ar
