"""CPU, 2 processes, gloo: W ranks x local batch == one process at the global batch (LayerNorm has no batch
statistics, dropout off): flat all-reduce-mean gradients and the AdamW step match the single-process run.
The ST blocks run through the emulated kernels, so this covers the real host path end to end."""
import os
import socket
import types

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make(seed=3, N=12):
    from stgcn_amd import models
    from tests.emu_util import bind_emulator, nonsym_gso
    bind_emulator()
    gso = torch.from_numpy(nonsym_gso(N, 2))
    args = types.SimpleNamespace(Kt=3, Ks=3, act_func="glu", graph_conv_type="cheb_graph_conv", gso=gso, enable_bias=True,
                                 droprate=0.0, n_his=12)
    torch.manual_seed(seed)
    return models.STGCNChebGraphConv(args, [[1], [64, 16, 64], [64, 16, 64], [128, 128], [1]], N)


def _data(N=12, B=4):
    rs = np.random.RandomState(0)
    return torch.from_numpy(rs.standard_normal((B, 1, 12, N))).float(), torch.from_numpy(rs.standard_normal((B, N))).float()


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from stgcn_amd.train import FlatGradAllReduce, init_distributed, make_optimizer, train_step
    torch.set_num_threads(1)
    r, _, w = init_distributed("gloo")
    assert (r, w) == (rank, world)
    model = _make()
    x, y = _data()
    bl = x.shape[0] // world
    xs, ys = x[rank * bl:(rank + 1) * bl], y[rank * bl:(rank + 1) * bl]
    opt = make_optimizer(model)
    ar = FlatGradAllReduce(list(model.parameters()), world)
    loss = train_step(model, opt, xs, ys, ar)
    # second step through the fused tail (what GraphedTrainStep captures for world > 1): deferred reductions flushed into the
    # flat GradArena, ONE all-reduce of that buffer, 1/world carried by the loss gradient, then AdamW
    from stgcn_amd.train import GradArena, fused_train_step
    arena = GradArena([p for p in model.parameters() if p.grad is not None])
    fused_train_step(model, opt, xs, ys, arena, world=world, all_reduce=dist.all_reduce)
    res = {"loss": float(loss), "grads": {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None},
           "params": {k: v.clone() for k, v in model.state_dict().items()}}
    if rank == 0:
        torch.save(res, out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_equal_one_big_batch(tmp_path):
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    from stgcn_amd.train import make_optimizer, train_step
    model = _make()
    x, y = _data()
    opt = make_optimizer(model)
    train_step(model, opt, x, y, None)
    train_step(model, opt, x, y, None)
    n_live = 0
    for k, p in model.named_parameters():
        if p.grad is None:
            assert k not in got["grads"]
            continue
        n_live += 1
        ref = p.grad
        assert (got["grads"][k] - ref).abs().max() <= 1e-5 * max(1.0, ref.abs().max().item()), k
    assert n_live == 28        # 38 tensors, 10 unused align convs never get a gradient
    for k, v in model.state_dict().items():
        assert (got["params"][k] - v).abs().max() <= 2e-6, k


def _tail_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from stgcn_amd.train import init_distributed, make_optimizer, tail_step
    torch.set_num_threads(1)
    init_distributed("gloo")
    model = _make()
    x, y = _data(B=3)                       # the partial last batch: 3 windows for 2 ranks -> shares of 2 and 1
    opt = make_optimizer(model)
    loss = tail_step(model, opt, x, y, world=world, rank=rank)
    if rank == 0:
        torch.save({"loss": float(loss), "params": {k: v.clone() for k, v in model.state_dict().items()}}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_tail_batch_weighted_by_local_count(tmp_path):
    """main.py:126-131 keeps the partial last batch: 2 ranks x (2 + 1 windows), loss gradients weighted by the local count and SUM
    all-reduced == one process stepping on the 3 windows."""
    out = str(tmp_path / "t0.pt")
    mp.spawn(_tail_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    from stgcn_amd.train import make_optimizer, tail_step
    model = _make()
    x, y = _data(B=3)
    opt = make_optimizer(model)
    loss = tail_step(model, opt, x, y)
    assert abs(got["loss"] - float(loss)) <= 1e-6 * max(1.0, abs(float(loss)))
    for k, v in model.state_dict().items():
        # (AdamW divides every gradient element by its own magnitude: an element whose gradient nearly cancels turns the different
        #  summation order of the two-rank run into a few 1e-6 of the lr = 1e-3 step)
        assert (got["params"][k] - v).abs().max() <= 1e-5, k


def _sync_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from stgcn_amd import models
    from stgcn_amd.train import init_distributed, sync_operators
    from tests.emu_util import bind_emulator, nonsym_gso
    torch.set_num_threads(1)
    init_distributed("gloo")
    bind_emulator()
    N = 12
    gso = torch.from_numpy(nonsym_gso(N, 2 + rank))      # every rank built ITS OWN operator (differently seeded numpy state)
    args = types.SimpleNamespace(Kt=3, Ks=3, act_func="glu", graph_conv_type="cheb_graph_conv", gso=gso, enable_bias=True, droprate=0.0, n_his=12)
    torch.manual_seed(3)
    model = models.STGCNChebGraphConv(args, [[1], [64, 16, 64], [64, 16, 64], [128, 128], [1]], N)
    x, _ = _data()
    before = model(x).detach().clone()
    sync_operators(model)
    blocks = [m for m in model.modules() if type(m).__name__ == "STConvBlock"]
    after = model(x).detach().clone()
    torch.save({"gso": [b.gso.clone() for b in blocks], "device": str(blocks[0].gso.device), "before": before, "after": after,
                "inner": [b.graph_conv.cheb_graph_conv.gso.clone() for b in blocks]}, out + f".{rank}")
    dist.barrier()
    dist.destroy_process_group()


def test_sync_operators_broadcasts_rank0_graph(tmp_path):
    """train.sync_operators had no test.  Two ranks that built different operators end up with rank 0's, on the device the
    operator lived on, in every place a block keeps it (block, graph-conv layer, inner graph conv), and the forward uses it."""
    from tests.emu_util import nonsym_gso
    out = str(tmp_path / "s")
    mp.spawn(_sync_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    g0 = torch.from_numpy(nonsym_gso(12, 2))
    for r in (r0, r1):
        assert r["device"] == "cpu"
        for g in r["gso"] + r["inner"]:
            assert torch.equal(g, g0)
    assert torch.equal(r0["before"], r0["after"])                       # rank 0 keeps its operator
    assert not torch.equal(r1["before"], r1["after"])                   # rank 1 trained on another graph before the broadcast
    assert torch.equal(r0["after"], r1["after"])                        # identical replicas afterwards


def _probe_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from stgcn_amd.train import init_distributed, probe_collective_capture
    init_distributed("gloo")
    res = probe_collective_capture(timeout_s=5.0)
    if rank == 0:
        torch.save(res, out)
    dist.barrier()
    dist.destroy_process_group()


def test_collective_capture_probe_declines_host_side_backends(tmp_path):
    """bench.py --gpus N asks train.probe_collective_capture() whether the all-reduce may be recorded inside the step's graph; a gloo group
    (CPU collectives, this test) must be turned down without starting the child experiment, and identically on every rank."""
    out = str(tmp_path / "p.pt")
    mp.spawn(_probe_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    ok, why = torch.load(out)
    assert ok is False and "gloo" in why


def test_capture_probe_child_environment():
    """The capture probe's child must host its own rendezvous store: the launcher's TORCHELASTIC_* variables (agent store) are dropped, the
    port is shifted off the parent group's, the rank variables survive."""
    from stgcn_amd.train import capture_child_env
    env = {"RANK": "3", "LOCAL_RANK": "3", "WORLD_SIZE": "8", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29400",
           "TORCHELASTIC_USE_AGENT_STORE": "True", "TORCHELASTIC_RUN_ID": "x", "HSA_ENABLE_IPC_MODE_LEGACY": "0"}
    c = capture_child_env(env)
    assert not any(k.startswith("TORCHELASTIC_") for k in c)
    assert c["MASTER_PORT"] == "29417" and c["RANK"] == "3" and c["WORLD_SIZE"] == "8" and c["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert env["MASTER_PORT"] == "29400" and "TORCHELASTIC_RUN_ID" in env      # (the caller's mapping is untouched)
