#!/usr/bin/env python3
"""Headline benchmark: training windows/sec, METR-LA ChebConv K=3 bs=32 fp32 (BASELINE.json configs[1]).

A "step" is the reference's loop body (main.py:165-169): zero_grad -> forward -> MSELoss -> backward ->
[gradient all-reduce] -> AdamW step, dropout p=0.5 ON, on one minibatch of 32 synthetic windows per GPU
(weak scaling: global batch = 32 * n_gpus, BASELINE.json configs[3] at 8 GPUs).  Inputs are resident in
HBM before the timed region.  Prints ONE JSON line (rank 0).

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, 64 FLOP/clk/SIMD @ 2.4 GHz
N_PRED = 12                         # C2 is "12 -> 12": n_his 12, n_pred 12
N_HIS, KT, KS, B_LOCAL = 12, 3, 3, int(os.environ.get("STGCN_BENCH_B", "32"))   # (env: batch-size sweeps of tools/, not the headline)
BLOCKS = [[1], [64, 16, 64], [64, 16, 64], [128, 128], [1]]


def load_gso(dataset="metr_la"):
    """Real METR-LA scaled Laplacian as main.py:97-101 builds it (committed fixture generated from the
    reference in the build container); falls back to a synthetic operator of the same size if missing."""
    path = os.path.join(ROOT, "tests", "golden", "gso_real.npz")
    if os.path.exists(path):
        return np.load(path)[dataset + ".cheb_sym_norm_lap"], "real METR-LA adj -> sym_norm_lap -> cheb GSO"
    rs = np.random.RandomState(0)
    a = rs.uniform(0.1, 1.0, (207, 207)) * (rs.uniform(size=(207, 207)) < 0.5)
    a = np.maximum(a, a.T)
    np.fill_diagonal(a, 1.0)
    d = 1.0 / np.sqrt(a.sum(1))
    lap = np.eye(207) - d[:, None] * a * d[None, :]
    return (2 * lap / np.linalg.eigvalsh(lap).max() - np.eye(207)).astype(np.float32), "synthetic 207-node graph"


def make_args(gso_t):
    import types
    return types.SimpleNamespace(Kt=KT, Ks=KS, act_func="glu", graph_conv_type="cheb_graph_conv", gso=gso_t,
                                 enable_bias=True, droprate=0.5, n_his=N_HIS)


def block_flops(B, c_in, T, N, need_dx):
    """Algorithmic FLOPs (2*MAC of conv/GEMM only) of one ST block by kernel label -- SURVEY.md section 8d closed form."""
    c0, c1, c2 = 64, 16, 64
    T1, T2 = T - KT + 1, T - 2 * (KT - 1)
    F_tc1 = 2 * B * T1 * N * KT * c_in * 2 * c0
    F_al = 2 * B * T1 * N * c0 * c1
    F_L = (KS - 1) * 2 * N * N * B * T1 * c1
    F_W = 2 * B * T1 * N * KS * c1 * c1
    F_tc2 = 2 * B * T2 * N * KT * c1 * 2 * c2
    # one entry per kernel label a launch of this path can carry (fused kernels and the stage-per-launch kernels they replace);
    # "_total" is the closed form of the whole block, independent of how the launches are cut
    return {"tconv_fwd.tc1": F_tc1 + F_al, "gconv_fwd": F_L + F_W, "tconv_fwd.tc2": F_tc2, "tc2_ln_fwd": F_tc2,
            "tconv_bwd_data.tc2": F_tc2, "gconv_bwd": F_L + 2 * F_W, "align_gate_bwd": 2 * F_al,
            "tconv_bwd_data.tc1": F_tc1 if need_dx else 0, "tconv_bwd_weight.tc1": F_tc1, "tconv_bwd_weight.tc2": F_tc2,
            "tc2_bwd": 2 * F_tc2, "tc1_bwd": (2 * F_tc1 if need_dx else F_tc1) + 2 * F_al,
            "_total": (F_tc1 + F_al + F_L + F_W + F_tc2) + (F_tc1 if need_dx else 0) + F_tc1 + 2 * (F_al + F_W + F_tc2) + F_L}


def stblock_flops_by_label(B, N):
    """{"<kernel label>@<block>": algorithmic FLOPs of that launch} for the two ST blocks of the C2 model."""
    tot = {}
    for blk, (c_in, T, need_dx) in enumerate(((1, N_HIS, False), (64, N_HIS - 2 * (KT - 1), True))):
        for k, v in block_flops(B, c_in, T, N, need_dx).items():
            tot[f"{k}@{blk}"] = v
    return tot


def cpu_baseline(gso_np, budget_s=20.0):
    """The reference's loop body restated by the oracle (torch CPU, all host cores), bounded sample."""
    from oracle import stgcn_oracle as orc
    # torch CPU intra-op threading collapses on very wide hosts (256 threads: 36 s/step measured in round 1),
    # so the baseline uses the best of a few thread counts; `cores` reports the count actually used.
    ncpu = os.cpu_count() or 1
    cfg = orc.OracleConfig(Kt=KT, Ks=KS, n_his=N_HIS, droprate=0.5, blocks=BLOCKS)
    N = gso_np.shape[0]
    p = orc.random_params(cfg, N, seed=0)
    gso = torch.from_numpy(gso_np)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B_LOCAL, 1, N_HIS, N, generator=g)
    y = torch.randn(B_LOCAL, N, generator=g)

    def masks():
        return [(torch.rand(B_LOCAL, 64, 8, N, generator=g) >= 0.5), (torch.rand(B_LOCAL, 64, 4, N, generator=g) >= 0.5),
                (torch.rand(B_LOCAL, 1, N, 128, generator=g) >= 0.5)]

    def one_step(state):
        t = time.perf_counter()
        orc.train_step(x, y, gso, p, cfg, state, keep_masks=masks())
        return time.perf_counter() - t

    best, cores = None, 1
    for th in sorted({min(ncpu, c) for c in (8, 16, 32, 64)}):
        torch.set_num_threads(th)
        st = {}
        one_step(st)                       # warm-up at this thread count
        dt = min(one_step(st), one_step(st))
        if best is None or dt < best:
            best, cores = dt, th
        if dt > 3.0:                       # hopeless configuration, stop probing wider
            break
    torch.set_num_threads(cores)
    state = {}
    one_step(state)
    n, t0 = 0, time.perf_counter()
    while True:
        one_step(state)
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 100:
            break
    return {"value": round(B_LOCAL * n / el, 2), "unit": "windows/s", "cores": cores, "kind": "port",
            "sample": f"{n} steps of bs {B_LOCAL} (C2 shapes, dropout on, AdamW) in {el:.1f} s, torch CPU oracle, {cores} of {ncpu} host threads"}


def pmc_traffic(label):
    """HBM-side bytes per launch of ``label`` from the newest committed PMC summary (profiles/*_pmc_traffic.json, written by
    tools/pmc_traffic.py from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same bench command:
    counters cannot be sampled from inside the process being timed).  None when no summary covers the kernel."""
    import glob
    for path in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*_pmc_traffic.json")), reverse=True):
        try:
            rec = json.load(open(path))["per_launch"].get(label)
        except (OSError, ValueError, KeyError):
            continue
        if rec:
            return int(rec["hbm_bytes"]), os.path.basename(path)
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="run the step eagerly instead of replaying a captured hipGraph")
    ap.add_argument("--no-resident-series", action="store_true",
                    help="feed (num, 1, n_his, N) window tensors copied per step instead of device-side windows of a resident series")
    ap.add_argument("--chains", type=int, default=int(os.environ.get("STGCN_CHAINS", "1")),
                    help="micro-batch chains of the minibatch run concurrently on separate HIP streams (train.chained_fwd_bwd)")
    args = ap.parse_args()

    from stgcn_amd import DropoutStream, _lib, models
    from stgcn_amd.train import FlatGradAllReduce, GraphedTrainStep, init_distributed, make_optimizer, train_step

    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (no CPU fallback)"
    rank, local_rank, world = init_distributed()
    assert world == args.gpus or world == 1 and args.gpus == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    L = _lib.lib()
    assert L.backend == "hip-gfx950"

    gso_np, gso_src = load_gso()
    N = gso_np.shape[0]
    torch.manual_seed(42)                       # identical replicas on every rank
    model = models.STGCNChebGraphConv(make_args(torch.from_numpy(gso_np).to(dev)), BLOCKS, N).to(dev)
    DropoutStream.manual_seed(1234 + rank)      # independent dropout streams per rank
    use_graph = not args.no_graph
    opt = make_optimizer(model, lr=1e-3, weight_decay=1e-3, capturable=use_graph)
    allreduce = FlatGradAllReduce(list(model.parameters()), world) if world > 1 else None

    # synthetic windows, resident in HBM: (num, 1, n_his, N) / (num, N) like script/dataloader.py:32-47
    n_batches = 16
    g = torch.Generator(device="cpu").manual_seed(7)
    x_all = torch.randn(n_batches * B_LOCAL * world, 1, N_HIS, N, generator=g).to(dev)
    y_all = torch.randn(n_batches * B_LOCAL * world, N, generator=g).to(dev)

    def batch(i):
        s = (i % n_batches) * B_LOCAL * world + rank * B_LOCAL
        return x_all[s:s + B_LOCAL], y_all[s:s + B_LOCAL]

    # the same amount of data as ONE resident (time, N) series (z-scored synthetic speeds): the captured step windows it on
    # the device (script/dataloader.py:32-47 without the 12x replicated tensor and without per-step input copies)
    resident = use_graph and not args.no_resident_series and args.chains == 1
    series = torch.randn(n_batches * B_LOCAL * world + N_HIS + N_PRED - 1, N, generator=g).to(dev) if resident else None

    model.train()
    step_i = 0
    graph_err = None
    if use_graph:
        try:
            graphed = GraphedTrainStep(model, opt, *batch(0), world=world, chains=args.chains, series=series, n_his=N_HIS, n_pred=N_PRED,
                                       rank=rank)

            def run_step(xb, yb):
                return graphed() if resident else graphed(xb, yb)
        except Exception as e:  # noqa: BLE001  -- same HIP path, just launched eagerly
            graph_err = repr(e)
            use_graph = False
            DropoutStream.disable_device_counter()
            torch.cuda.synchronize()
            opt = make_optimizer(model, lr=1e-3, weight_decay=1e-3, capturable=False)
        if world > 1:   # every rank must take the same path
            flag = torch.tensor([1 if use_graph else 0], device=dev)
            torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
            if use_graph and int(flag.item()) == 0:
                use_graph, graph_err = False, "another rank failed to capture"
                DropoutStream.disable_device_counter()
                opt = make_optimizer(model, lr=1e-3, weight_decay=1e-3, capturable=False)
    if not use_graph:
        def run_step(xb, yb):
            return train_step(model, opt, xb, yb, allreduce)
    for _ in range(args.warmup):
        run_step(*batch(step_i))
        step_i += 1

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = run_step(*batch(step_i))
        step_i += 1
    sync()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        el = float(t.item())
    loss_val = float(loss.item())
    assert np.isfinite(loss_val), "training diverged"

    out = {"metric": "training windows/sec, METR-LA ChebConv K=3 bs=32", "value": round(B_LOCAL * world * args.steps / el, 2),
           "unit": "windows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(1e3 * el / args.steps, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f32", "data": "synthetic",
           "config": {"workload": "C2: METR-LA 207 nodes, STGCNChebGraphConv Ks=3 Kt=3, n_his=12, bs=32 per GPU, fp32, "
                                  "dropout 0.5, AdamW lr 1e-3 wd 1e-3; full step zero_grad+fwd+MSE+bwd+opt",
                      "graph": gso_src, "global_batch": B_LOCAL * world, "parallelism": f"dp{world}",
                      "output_block": "fused HIP path (stgcn_outblock_*)", "final_loss": round(loss_val, 5),
                      "launch": "hipGraph replay" if use_graph else "eager", "graph_error": graph_err,
                      "chains": args.chains if use_graph else 1,
                      "input": ("device-side windows (n_his 12, n_pred 12) of a resident (time, N) series, batch position on the device"
                                if (resident and use_graph) else "(num, 1, n_his, N) window tensors, one batch copied per step")}}

    if rank == 0 and not args.no_profile:
        # per-kernel durations with hipEvents on the launch stream, over the same K steps (second pass)
        # (eager launches: hipEvents cannot be recorded inside a graph replay; the kernels and shapes are the same)
        DropoutStream.disable_device_counter()
        L.dll.stgcn_profile_enable(1)
        ksteps = min(args.steps, 50)
        for _ in range(ksteps):
            train_step(model, opt, *batch(step_i), None)
            step_i += 1
        torch.cuda.synchronize()
        buf = C.create_string_buffer(1 << 14)
        L.check(L.dll.stgcn_profile_collect(buf, len(buf)), "stgcn_profile_collect")
        L.dll.stgcn_profile_enable(0)
        prof = json.loads(buf.value.decode())
        flops = stblock_flops_by_label(B_LOCAL, N)
        per_step = {k: v["total_ms"] / ksteps for k, v in prof.items()}
        stblock_total = sum(v for k, v in flops.items() if k.startswith("_total"))
        mfma_kernels = {k: per_step[k] for k in flops if k in per_step and flops[k] > 0}
        dom = max(mfma_kernels, key=mfma_kernels.get)          # the single launch (kernel @ block) that costs most
        calls_per_step = prof[dom]["calls"] / ksteps
        dur_ms = prof[dom]["total_ms"] / prof[dom]["calls"]
        ach = flops[dom] / calls_per_step / (dur_ms * 1e-3) / 1e12
        tot_ms = sum(v for k, v in per_step.items() if not k.startswith(("head.", "adamw")))
        traffic, traffic_src = pmc_traffic(dom)
        out["roofline"] = {"bound": "mfma", "kernel": dom, "achieved": round(ach, 3), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                           "frac": round(ach / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": traffic, "traffic_unit": "bytes/launch",
                           "traffic_source": traffic_src,
                           "avg_launch_us": round(dur_ms * 1e3, 2), "flops_per_launch": int(flops[dom] / calls_per_step),
                           "stblock_kernels_ms_per_step": round(tot_ms, 4), "all_kernels_ms_per_step": round(sum(per_step.values()), 4),
                           "stblock_fwd_bwd_frac": round(stblock_total / (tot_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                           "stblock_flops_per_step": int(stblock_total),
                           "stblock_launches_per_step": int(round(sum(v["calls"] for k, v in prof.items()
                                                                        if not k.startswith(("head.", "adamw", "prepack", "mse", "reduce"))) / ksteps)),
                           "per_kernel_us_per_step": {k: round(v * 1e3, 2) for k, v in sorted(per_step.items())}}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(gso_np)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
