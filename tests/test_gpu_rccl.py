"""-m gpu, needs >= 2 GPUs (skipped on the 1-GPU test box): the data-parallel captured step with ONE rank per GPU exchanging through
RCCL (torch.distributed backend "nccl") == one process at the global batch.  Same scenario as tests/test_gpu_dp.py, which runs the
two ranks on one GPU over gloo; here the collective itself (RCCL all-reduce of the flat gradient arena over xGMI) is under test."""
import os
import types

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.test_gpu_dp import BG, N_HIS, N_PRED, _free_port, _series, _windows

pytestmark = pytest.mark.gpu


def _make(dev):
    from stgcn_amd import models
    from tests.gpu_util import bind_hip
    from tests.helpers import real_gso
    bind_hip()
    gso = torch.from_numpy(real_gso("metr_la.cheb_sym_norm_lap")).to(dev)
    args = types.SimpleNamespace(Kt=3, Ks=3, act_func="glu", graph_conv_type="cheb_graph_conv", gso=gso, enable_bias=True,
                                 droprate=0.0, n_his=12)
    torch.manual_seed(1)
    return models.STGCNChebGraphConv(args, [[1], [64, 16, 64], [64, 16, 64], [128, 128], [1]], 207).to(dev)


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    from stgcn_amd.train import GraphedTrainStep, init_distributed, make_optimizer
    r, lr, w = init_distributed("nccl")
    assert (r, w) == (rank, world) and torch.cuda.current_device() == rank
    dev = torch.device("cuda", rank)
    model = _make(dev)
    opt = make_optimizer(model, capturable=True)
    series = _series().to(dev)
    bl = BG // world
    x0, y0 = _windows(series, rank * bl, bl)
    gs = GraphedTrainStep(model, opt, x0, y0, world=world, warmup=2, series=series, n_his=N_HIS, n_pred=N_PRED, rank=rank)
    for _ in range(3):
        gs()
    torch.cuda.synchronize(dev)
    if rank == 0:
        torch.save({k: v.cpu() for k, v in model.state_dict().items()}, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL all-reduce over xGMI)")
def test_two_gpus_rccl_equal_one_big_batch(tmp_path):
    from stgcn_amd import DropoutStream
    from stgcn_amd.train import make_optimizer, train_step
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    DropoutStream.disable_device_counter()
    model = _make("cuda:0")
    opt = make_optimizer(model)
    series = _series().cuda()
    for k in range(6):      # constructor: global steps 0, 1 (warm-up) and 2 (verification replay), then 3 replays
        train_step(model, opt, *_windows(series, k * BG, BG))
    torch.cuda.synchronize()
    diffs = torch.cat([(got[k] - v.cpu()).abs().flatten() for k, v in model.state_dict().items()])
    errs = {k: float((got[k] - v.cpu()).abs().max()) for k, v in model.state_dict().items()}
    assert float(torch.quantile(diffs[torch.randperm(diffs.numel())[:200000]], 0.999)) <= 2e-5, sorted(errs.items(), key=lambda kv: -kv[1])[:4]
    assert max(errs.values()) <= 5e-4, sorted(errs.items(), key=lambda kv: -kv[1])[:4]


def test_rccl_all_reduce_inside_a_hipgraph_single_rank():
    """The experiment `train.probe_collective_capture` runs on every rank before bench.py --gpus N records the gradient all-reduce inside
    the step's graph, here with ONE rank on the one GPU of the test box: an RCCL all-reduce is captured, replayed three
    times and its result checked, in a child process under the watchdog.  (Two ranks cannot share a GPU under RCCL; the two-rank form of the
    same code runs on the driver's multi-GPU node.)"""
    from stgcn_amd.train import run_collective_capture_child
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    ok, why = run_collective_capture_child(env, timeout_s=120.0)
    assert ok, why
