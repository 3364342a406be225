#!/usr/bin/env python3
"""Headline benchmark: training windows/sec, METR-LA ChebConv K=3 bs=32 fp32 (BASELINE.json configs[1]).

A "step" is the reference's loop body (main.py:165-169): zero_grad -> forward -> MSELoss -> backward ->
[gradient all-reduce] -> AdamW step, dropout p=0.5 ON, on one minibatch of synthetic windows per GPU
(weak scaling: global batch = 32 * n_gpus, BASELINE.json configs[3] at 8 GPUs).  Inputs are resident in
HBM before the timed region.  Prints ONE JSON line (rank 0).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c3|c5] [--dtype f32|bf16]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

--config (default c2 = the headline; the others are the remaining single-GPU entries of BASELINE.json `configs`):
  c2  METR-LA, 207 nodes, ChebConv Ks=3, bs 32, 12->12, fp32                      (configs[1] / [3])
  c3  PEMS-BAY, 325 nodes, ChebConv Ks=3, bs 64                                   (configs[2]; quoted in bf16: default --dtype bf16)
  c5  synthetic dense 8192-node graph, ChebConv Ks=5, bs 16, tiled graph conv     (configs[4]; quoted in bf16: default --dtype bf16;
                                                                                    with --dtype f32, --gc-precision selects the operator products)
--dtype: storage / arithmetic type of the activations (default: what BASELINE.json quotes the config in).  bf16 = bf16 activations and
saved tensors, bf16 matrix cores with fp32 accumulation; parameters, LayerNorm statistics, gradients, AdamW stay fp32.
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
PEAK_BF16_MFMA_TFLOPS = 2500.0     # MI355X_MICROARCH.md: dense bf16 (~2.5 PF)
PEAK_HBM_GBS = 8000.0              # MI355X_MICROARCH.md: HBM3E ~8 TB/s
N_HIS, KT = 12, 3
BLOCKS = [[1], [64, 16, 64], [64, 16, 64], [128, 128], [1]]
CONFIGS = {
    "c2": dict(N=207, Ks=3, B=32, n_pred=12, gso="metr_la.cheb_sym_norm_lap", metric="training windows/sec, METR-LA ChebConv K=3 bs=32",
               dtype="f32", compulsory_bytes={"f32": 82_000_000, "bf16": 41_000_000},
               workload="C2: METR-LA 207 nodes, STGCNChebGraphConv Ks=3 Kt=3, n_his=12, bs=32 per GPU, {dtype}, dropout 0.5, AdamW lr 1e-3 wd 1e-3; "
                        "full step zero_grad+fwd+MSE+bwd+opt"),
    "c3": dict(N=325, Ks=3, B=64, n_pred=12, gso="pems_bay.cheb_sym_norm_lap", metric="training windows/sec, PEMS-BAY ChebConv K=3 bs=64",
               dtype="bf16", compulsory_bytes={"bf16": 129_000_000, "f32": 258_000_000},
               workload="C3: PEMS-BAY 325 nodes, STGCNChebGraphConv Ks=3 Kt=3, n_his=12, bs=64, {dtype} activations, dropout 0.5, AdamW; full step"),
    "c5": dict(N=8192, Ks=5, B=16, n_pred=12, gso=None, metric="training windows/sec, synthetic 8192-node dense graph ChebConv K=5 bs=16",
               dtype="bf16", compulsory_bytes={"bf16": 1_350_000_000, "f32": 2_700_000_000},
               workload="C5: synthetic dense 8192-node graph, STGCNChebGraphConv Ks=5 Kt=3, n_his=12, bs=16, {dtype} activations, tiled graph conv, "
                        "dropout 0.5, AdamW; full step"),
}
# (named explicitly: they have to be re-measured whenever a kernel's traffic changes; keyed by (config, dtype))
PMC_TRAFFIC_FILES = {("c2", "f32"): "r6-60_pmc_traffic.json", ("c3", "bf16"): "r6-60_pmc_traffic_c3_bf16.json",
                     ("c5", "bf16"): "r6-60_pmc_traffic_c5_bf16.json"}
B_OVERRIDE = os.environ.get("STGCN_BENCH_B")       # (env: batch-size sweeps of tools/, not the headline)


def load_gso(cfg):
    """The real scaled Laplacian as main.py:97-101 builds it (committed fixture generated from the reference in the build
    container); c5: a synthetic dense symmetric operator with infinity norm 1 (only its dense N x N shape matters for timing)."""
    N = cfg["N"]
    path = os.path.join(ROOT, "tests", "golden", "gso_real.npz")
    if cfg["gso"] and os.path.exists(path):
        return np.load(path)[cfg["gso"]], f"real {cfg['gso'].split('.')[0]} adj -> sym_norm_lap -> cheb GSO"
    rs = np.random.RandomState(0)
    a = (rs.uniform(0.1, 1.0, (N, N)) * (rs.uniform(size=(N, N)) < 0.4)).astype(np.float32)
    a = np.maximum(a, a.T)
    a /= a.sum(1).max()
    return a, f"synthetic dense symmetric {N}-node operator (40 % dense, infinity norm 1)"


def make_args(gso_t, Ks):
    import types
    return types.SimpleNamespace(Kt=KT, Ks=Ks, act_func="glu", graph_conv_type="cheb_graph_conv", gso=gso_t,
                                 enable_bias=True, droprate=0.5, n_his=N_HIS)


def block_flops(B, c_in, T, N, Ks, need_dx):
    """Algorithmic FLOPs (2*MAC of conv/GEMM only) of one ST block by kernel label -- SURVEY.md section 8d closed form."""
    c0, c1, c2 = 64, 16, 64
    T1, T2 = T - KT + 1, T - 2 * (KT - 1)
    F_tc1 = 2 * B * T1 * N * KT * c_in * 2 * c0
    F_al = 2 * B * T1 * N * c0 * c1
    F_L = (Ks - 1) * 2 * N * N * B * T1 * c1
    F_W = 2 * B * T1 * N * Ks * c1 * c1
    F_tc2 = 2 * B * T2 * N * KT * c1 * 2 * c2
    # one entry per kernel label a launch of this path can carry (fused kernels and the stage-per-launch kernels they replace);
    # "_total" is the closed form of the whole block, independent of how the launches are cut
    return {"tconv_fwd.tc1": F_tc1 + F_al, "gconv_fwd": F_L + F_W, "tconv_fwd.tc2": F_tc2, "tc2_ln_fwd": F_tc2,
            "tconv_bwd_data.tc2": F_tc2, "gconv_bwd": F_L + 2 * F_W, "align_gate_bwd": 2 * F_al,
            "tconv_bwd_data.tc1": F_tc1 if need_dx else 0, "tconv_bwd_weight.tc1": F_tc1, "tconv_bwd_weight.tc2": F_tc2,
            "tc2_bwd": 2 * F_tc2, "tc1_bwd": (2 * F_tc1 if need_dx else F_tc1) + 2 * F_al,
            # chained launches: several stages as roles of one launch
            "stblock_fwd": F_tc1 + F_al + F_L + F_W + F_tc2, "tc1_gconv_fwd": F_tc1 + F_al + F_L + F_W,
            "gso_gemm_fwd": F_L / max(Ks - 1, 1), "gso_gemm_bwd": F_L / max(Ks - 1, 1),
            "_total": (F_tc1 + F_al + F_L + F_W + F_tc2) + (F_tc1 if need_dx else 0) + F_tc1 + 2 * (F_al + F_W + F_tc2) + F_L}


def block_bytes(B, c_in, T, N, Ks, need_dx, e, part):
    """Algorithmic HBM bytes of ONE launch by kernel label: every tensor the kernel must read or write once (activation elements x
    element size e; fp32 LayerNorm parameters / weight-gradient partials `part[label]` in floats), no re-reads -- the figure
    `roofline.achieved` is priced with when a kernel is HBM-bound (SURVEY.md section 8d convention, per kernel instead of per block)."""
    c0, c1, c2 = 64, 16, 64
    T1, T2 = T - KT + 1, T - 2 * (KT - 1)
    r0, r1, r2 = B * T * N, B * T1 * N, B * T2 * N
    ln = 2 * N * c2 * 4
    recompute = KT * c_in <= 16
    tc1b = e * (r0 * c_in + (0 if recompute else 2 * r1 * c0) + r1 * c1)
    return {"tconv_fwd.tc1": tc1b,
            # chained launches: the hand-off tensors (A, G) are written once (they are saved for backward) and re-read on chip / from L2
            "tc1_gconv_fwd": tc1b + e * (r1 * c1 * Ks), "stblock_fwd": tc1b + e * (r1 * c1 * Ks) + e * (3 * r2 * c2) + ln,
            "gconv_fwd": e * (r1 * c1 * (Ks + 1)),
            "tc2_ln_fwd": e * (r1 * c1 + 3 * r2 * c2) + ln,
            "tc2_bwd": e * (3 * r2 * c2 + 2 * r1 * c1) + ln // 2 + 4 * part.get("tc2_bwd", 0),
            "gconv_bwd": e * (r1 * c1 * (Ks + 2)) + 4 * part.get("gconv_bwd", 0),
            "tc1_bwd": e * (r1 * c1 + 2 * r1 * c0 + (2 if need_dx else 1) * r0 * c_in + (2 * r0 * c_in if need_dx else 0)) + 4 * part.get("tc1_bwd", 0),
            "align_gate_bwd": e * (r1 * c1 + r0 * c_in) + 4 * part.get("align_gate_bwd", 0)}


def stblock_bytes_by_label(B, N, Ks, e):
    tot = {}
    nt = (N + 15) // 16
    for blk, (c_in, T, need_dx) in enumerate(((1, N_HIS, False), (64, N_HIS - 2 * (KT - 1), True))):
        T1 = T - KT + 1
        part = {"tc2_bwd": min(B * nt, 512) * (KT * 16 * 128 + 128) + 2 * B * N * 64, "gconv_bwd": B * T1 * (Ks + 1) * 256,
                "tc1_bwd": 256 * (KT * c_in * 128 + 128 + 64 * 16 + 16), "align_gate_bwd": 512 * (64 * 16 + 16 + 16 * 128 + 128)}
        for k, v in block_bytes(B, c_in, T, N, Ks, need_dx, e, part).items():
            tot[f"{k}@{blk}"] = v
    return tot


def stblock_flops_by_label(B, N, Ks=3):
    """{"<kernel label>@<block>": algorithmic FLOPs of ONE launch with that label} for the two ST blocks of the model."""
    tot = {}
    for blk, (c_in, T, need_dx) in enumerate(((1, N_HIS, False), (64, N_HIS - 2 * (KT - 1), True))):
        for k, v in block_flops(B, c_in, T, N, Ks, need_dx).items():
            tot[f"{k}@{blk}"] = v
    return tot


def cpu_baseline(gso_np, cfg, B, budget_s=20.0, probe=True):
    """The reference's loop body restated by the oracle (torch CPU), bounded sample.  kind = "port": the reference's own modules
    live in /root/reference, which does not exist on the GPU box; the oracle is the line-by-line restatement pinned against them
    (tests/test_oracle_golden.py), with the dropout masks drawn up front (the reference's nn.Dropout spends 36 % of its CPU step in
    bernoulli_, SURVEY.md section 3.3 -- this port is therefore a FASTER CPU baseline than the reference itself)."""
    from oracle import stgcn_oracle as orc
    # torch CPU intra-op threading collapses on very wide hosts (256 threads: 36 s/step measured in round 1),
    # so the baseline uses the best of a few thread counts; `cores` reports the count actually used.
    ncpu = os.cpu_count() or 1
    ocfg = orc.OracleConfig(Kt=KT, Ks=cfg["Ks"], n_his=N_HIS, droprate=0.5, blocks=BLOCKS)
    N = gso_np.shape[0]
    p = orc.random_params(ocfg, N, seed=0)
    gso = torch.from_numpy(gso_np)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, 1, N_HIS, N, generator=g)
    y = torch.randn(B, N, generator=g)

    def masks():
        return [(torch.rand(B, 64, 8, N, generator=g) >= 0.5), (torch.rand(B, 64, 4, N, generator=g) >= 0.5),
                (torch.rand(B, 1, N, 128, generator=g) >= 0.5)]

    def one_step(state):
        t = time.perf_counter()
        orc.train_step(x, y, gso, p, ocfg, state, keep_masks=masks())
        return time.perf_counter() - t

    best, cores = None, min(ncpu, 32)
    for th in (sorted({min(ncpu, c) for c in (8, 16, 32, 64)}) if probe else []):
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
    if probe:
        one_step(state)
    n, t0 = 0, time.perf_counter()
    while True:
        one_step(state)
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 100:
            break
    return {"value": round(B * n / el, 2), "unit": "windows/s", "cores": cores, "kind": "port",
            "sample": f"{n} steps of bs {B} ({cfg['workload'][:2]} shapes, dropout on (masks pre-drawn), AdamW) in {el:.1f} s, torch CPU oracle "
                      f"(restatement of the reference modules), {cores} of {ncpu} host threads" + ("" if probe else "; no warm-up step, thread count not probed")}


def side_config(name, steps, warmup, timeout_s=420):
    """One of the other single-GPU entries of BASELINE.json `configs` (c3: configs[2], c5: configs[4]) measured by THIS script in a fresh
    process (its own library modes, model and captured graph), summarised for the headline line: value, step time, dtype, the dominant
    launch against its roofline, the CPU baseline of the same shapes."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--config", name, "--steps", str(steps), "--warmup", str(warmup), "--no-gpu-baseline",
           "--no-side-configs", "--cpu-budget", "10"]
    t0 = time.perf_counter()
    try:
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, cwd=ROOT)
        rec = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    except Exception as e:  # noqa: BLE001  (a failing side config must not lose the headline line)
        return {"error": repr(e)[:300]}
    rl = rec.get("roofline") or {}
    return {"metric": rec["metric"], "value": rec["value"], "unit": rec["unit"], "ms_per_step": rec["ms_per_step"], "steps": rec["steps"],
            "dtype": rec["dtype"], "workload": rec["config"]["workload"], "launch": rec["config"]["launch"], "final_loss": rec["config"]["final_loss"],
            "roofline": {k: rl.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "avg_launch_us", "stblock_kernels_ms_per_step",
                                               "stblock_fwd_bwd_frac", "stblock_hbm_frac", "stblock_traffic_bytes", "stblock_compulsory_bytes")},
            "cpu_baseline": rec.get("cpu_baseline"), "wall_s": round(time.perf_counter() - t0, 1)}


def gpu_baseline(model, gso_t, cfg, B, N, dev):
    """SURVEY.md section 8d "second reported baseline": the same parameters through stock PyTorch-ROCm ops (MIOpen conv2d, rocBLAS
    einsum, ATen LayerNorm / dropout / AdamW) on the same GPU, the reference's loop body, eager launches."""
    from tools.torch_baseline import time_train_step
    g = torch.Generator(device="cpu").manual_seed(11)
    x = torch.randn(B, 1, N_HIS, N, generator=g).to(dev)
    y = torch.randn(B, N, generator=g).to(dev)
    steps = 20 if N <= 1024 else 3
    ms, _ = time_train_step({k: v.detach().clone() for k, v in model.state_dict().items()}, gso_t, x, y, Kt=KT, Ks=cfg["Ks"], n_his=N_HIS,
                            blocks=BLOCKS, droprate=0.5, steps=steps, warmup=3)
    return {"value": round(B / (ms * 1e-3), 2), "unit": "windows/s", "ms_per_step": round(ms, 4), "steps": steps,
            "kind": "same parameters and step through stock PyTorch-ROCm ops (MIOpen / rocBLAS / ATen), eager, same GPU"}


def pmc_traffic(config, dtype):
    """HBM-side bytes per launch from the committed PMC summary of the current kernels (profiles/PMC_TRAFFIC_FILES[...], written by tools/pmc_traffic.py
    from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same bench command: counters cannot be sampled from inside
    the process being timed).  Returns ({label: bytes}, file name) or ({}, None)."""
    name = PMC_TRAFFIC_FILES.get((config, dtype))
    if name is None:
        return {}, None
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", name)))["per_launch"]
    except (OSError, ValueError, KeyError):
        return {}, None
    return {k: int(v["hbm_bytes"]) for k, v in rec.items()}, name


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--dtype", default=None, choices=["f32", "bf16"], help="activation storage / arithmetic (default: the config's own: c2 f32, c3 / c5 bf16)")
    ap.add_argument("--gc-precision", default="fp32", choices=["fp32", "bf16x3", "bf16"], help="operator products of the graph conv: tiled path (c5) all modes; slab path (c2, c3) fp32 or bf16x3 (forward only, opt-in)")
    ap.add_argument("--bwd-precision", default="fp32", choices=["fp32", "bf16x3"],
                    help="matrix products of the backward kernels of fp32 blocks: exact fp32 MFMAs (default, the headline) or split-bf16 operands "
                         "(three bf16 MFMAs per product, ~2^-16 relative: inside the 1e-3 gradient bar, NOT fp32 arithmetic -- an opt-in mode)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary measurement of the bf16x3-backward mode")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-side-configs", action="store_true", help="c2 only: do not append the c3 / c5 (bf16) measurements as `side_configs`")
    ap.add_argument("--cpu-budget", type=float, default=20.0, help="seconds of timed CPU-baseline steps")
    ap.add_argument("--replay-times", action="store_true", help="diagnostic: config.replay_ms = GPU time of each of the first steps after a synchronisation")
    ap.add_argument("--no-graph", action="store_true", help="run the step eagerly instead of replaying a captured hipGraph")
    ap.add_argument("--no-resident-series", action="store_true",
                    help="feed (num, 1, n_his, N) window tensors copied per step instead of device-side windows of a resident series")
    ap.add_argument("--capture-collective", default="auto", choices=["auto", "off"],
                    help="N > 1: record the gradient all-reduce inside the step's hipGraph (one graph per step) when a watchdogged probe in a child "
                         "process shows that RCCL capture works on this machine (auto), or keep the two-graph form around an eager all-reduce (off)")
    ap.add_argument("--chains", type=int, default=int(os.environ.get("STGCN_CHAINS", "1")),
                    help="micro-batch chains of the minibatch run concurrently on separate HIP streams (train.chained_fwd_bwd)")
    args = ap.parse_args()
    cfg = dict(CONFIGS[args.config])
    DTYPE = args.dtype or cfg["dtype"]
    cfg["workload"] = cfg["workload"].format(dtype="fp32" if DTYPE == "f32" else "bf16")
    B_LOCAL = int(B_OVERRIDE) if B_OVERRIDE else cfg["B"]
    N_PRED, KS = cfg["n_pred"], cfg["Ks"]

    from stgcn_amd import DropoutStream, _lib, models, ops
    from stgcn_amd.train import FlatGradAllReduce, GraphedTrainStep, init_distributed, make_optimizer, train_step

    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (no CPU fallback)"
    # (test hook: STGCN_BENCH_BACKEND=gloo + STGCN_BENCH_SHARE_GPU=1 drive the N-rank code path with every rank on GPU 0 of a
    #  one-GPU box -- the numbers of such a run mean nothing, the control flow is what is being exercised)
    share_gpu = os.environ.get("STGCN_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        os.environ["LOCAL_RANK"] = "0"
    rank, local_rank, world = init_distributed(os.environ.get("STGCN_BENCH_BACKEND"))
    assert world == args.gpus or world == 1 and args.gpus == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    L = _lib.lib()
    assert L.backend == "hip-gfx950"
    if cfg["N"] > 512:
        ops.set_gc_precision(args.gc_precision)
    elif args.gc_precision == "bf16x3":
        ops.set_slab_gc_precision("bf16x3")          # (opt-in: forward operator products of the slab-resident graph conv)

    ops.set_bwd_precision(args.bwd_precision)
    if os.environ.get("STGCN_BENCH_TC2LN_PP"):      # (env: A/B runs of tools/gpu_ab.sh, not the headline -- default: the library's own rule)
        ops.set_tc2ln_peers(int(os.environ["STGCN_BENCH_TC2LN_PP"]))
    gso_np, gso_src = load_gso(cfg)
    N = gso_np.shape[0]
    gso_t = torch.from_numpy(gso_np).to(dev)
    torch.manual_seed(42)                       # identical replicas on every rank
    model = models.STGCNChebGraphConv(make_args(gso_t, KS), BLOCKS, N).to(dev)
    if DTYPE == "bf16":
        model.set_compute_dtype(torch.bfloat16)
    DropoutStream.manual_seed(1234 + rank)      # independent dropout streams per rank
    use_graph = not args.no_graph
    opt = make_optimizer(model, lr=1e-3, weight_decay=1e-3, capturable=use_graph)
    allreduce = FlatGradAllReduce(list(model.parameters()), world) if world > 1 else None

    # synthetic windows, resident in HBM: (num, 1, n_his, N) / (num, N) like script/dataloader.py:32-47
    n_batches = 16 if N <= 1024 else 2
    g = torch.Generator(device="cpu").manual_seed(7)
    x_all = torch.randn(n_batches * B_LOCAL * world, 1, N_HIS, N, generator=g).to(dev)
    y_all = torch.randn(n_batches * B_LOCAL * world, N, generator=g).to(dev)

    def batch(i):
        s = (i % n_batches) * B_LOCAL * world + rank * B_LOCAL
        return x_all[s:s + B_LOCAL], y_all[s:s + B_LOCAL]

    # the same amount of data as ONE resident (time, N) series (z-scored synthetic speeds): the captured step windows it on
    # the device (script/dataloader.py:32-47 without the 12x replicated tensor and without per-step input copies)
    resident = use_graph and not args.no_resident_series and args.chains == 1
    series = torch.randn(n_batches * B_LOCAL * world + N_HIS + N_PRED, N, generator=g).to(dev) if resident else None

    model.train()
    step_i = 0
    graph_err = None
    graphed = None
    capture_coll, coll_why = False, None
    if use_graph and world > 1 and args.capture_collective == "auto":
        from stgcn_amd.train import probe_collective_capture
        capture_coll, coll_why = probe_collective_capture(timeout_s=60.0)      # (every rank: the verdict is agreed on inside)
    if use_graph:
        try:
            try:
                graphed = GraphedTrainStep(model, opt, *batch(0), world=world, chains=args.chains, series=series, n_his=N_HIS, n_pred=N_PRED,
                                           rank=rank, capture_collective=capture_coll)
            except Exception as e:  # noqa: BLE001  -- the probe passed but the real capture did not: the two-graph form is the known-good one
                if not capture_coll:
                    raise
                capture_coll, coll_why = False, "captured step failed: " + repr(e)[:200]
                torch.cuda.synchronize()
                graphed = GraphedTrainStep(model, opt, *batch(0), world=world, chains=args.chains, series=series, n_his=N_HIS, n_pred=N_PRED,
                                           rank=rank, capture_collective=False)

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
    from stgcn_amd.train import check_in_launch_waits
    check_in_launch_waits(model)      # (after the timed region: a starved in-launch wait would have made the loss NaN -- this says where)
    assert np.isfinite(loss_val), "training diverged"
    replay_ms = None
    if args.replay_times:   # diagnostic (after the timed region): the GPU time of each of the first steps after a synchronisation
        n_ev = min(args.steps, 24)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(n_ev + 1)]
        sync()
        tw0 = time.perf_counter()
        for i in range(n_ev):
            ev[i].record()
            run_step(*batch(step_i))
            step_i += 1
        ev[n_ev].record()
        tw1 = time.perf_counter()
        sync()
        tw2 = time.perf_counter()
        replay_ms = {"gpu_ms_per_step": [round(ev[i].elapsed_time(ev[i + 1]), 4) for i in range(n_ev)],
                     "host_enqueue_ms": round(1e3 * (tw1 - tw0), 3), "host_total_ms": round(1e3 * (tw2 - tw0), 3)}
    n_gpus = torch.distributed.get_world_size() if (world > 1 and torch.distributed.is_initialized()) else 1

    out = {"metric": cfg["metric"], "value": round(B_LOCAL * world * args.steps / el, 2),
           "unit": "windows/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(1e3 * el / args.steps, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": DTYPE, "data": "synthetic",
           "config": {"workload": cfg["workload"], "graph": gso_src, "global_batch": B_LOCAL * world, "parallelism": f"dp{world}",
                      "output_block": "fused HIP path (stgcn_outblock_*)", "final_loss": round(loss_val, 5),
                      "launch": "hipGraph replay" if use_graph else "eager", "graph_error": graph_err,
                      **({"replay_ms": replay_ms} if replay_ms else {}),
                      "chains": args.chains if use_graph else 1,
                      "backward_products": ("bf16" if DTYPE == "bf16" else args.bwd_precision),
                      **({"matrix_products": ("fp32 in, fp32 accumulate; tc1_fwd / tc2_ln_fwd / tc1_bwd form their products as bf16x6 -- both operands split EXACTLY into "
                                              "three bf16 (8+8+8 significand bits), six of the nine partial products (the others < 2^-32) on v_mfma_f32_16x16x32_bf16; error "
                                              "against the fp64 stage oracle equal to v_mfma_f32_16x16x4_f32's (profiles/r6-60_x6_errors.txt); every other product on "
                                              "v_mfma_f32_16x16x4_f32" if os.environ.get("STGCN_MFMA_X6", "1") != "0" else "v_mfma_f32_16x16x4_f32 for every product (STGCN_MFMA_X6=0)")}
                         if DTYPE == "f32" else {}),
                      "operator_products": ("bf16" if DTYPE == "bf16" else args.gc_precision if (N > 512 or args.gc_precision == "bf16x3") else "fp32"),
                      "input": (f"device-side windows (n_his 12, n_pred {N_PRED}) of a resident (time, N) series, batch position on the device"
                                if (resident and use_graph) else "(num, 1, n_his, N) window tensors, one batch copied per step")}}

    if world > 1:
        # the step's one collective, timed on its own: ~1 MB flat fp32 gradient buffer, latency-bound on xGMI
        flat = graphed.flat if (use_graph and graphed is not None and graphed.flat is not None) else torch.zeros(230000, device=dev)
        for _ in range(5):
            torch.distributed.all_reduce(flat)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            torch.distributed.all_reduce(flat)
        e1.record()
        torch.cuda.synchronize()
        ar_us = 1e3 * e0.elapsed_time(e1) / 50
        in_graph = bool(use_graph and graphed is not None and graphed.capture_collective)
        # what the collective really costs the step (measured, not assumed): the two-graph form replayed WITHOUT the
        # all-reduce between its graphs (gradients stay rank-local: timing only, after the headline measurement), against the step itself.
        # The one-graph form cannot drop its collective; there the figure stays null.
        exposed_us = None
        if use_graph and graphed is not None and graphed.g2 is not None:
            torch.distributed.barrier()
            torch.cuda.synchronize()
            e0.record()
            for _ in range(50):
                graphed.g1.replay()
                graphed.g2.replay()
            e1.record()
            torch.cuda.synchronize()
            exposed_us = round(1e6 * el / args.steps - 1e3 * e0.elapsed_time(e1) / 50, 2)
            torch.distributed.barrier()
        out["config"]["allreduce"] = {"bytes": int(flat.numel() * 4), "us": round(ar_us, 2), "exposed_us": exposed_us,
                                      # nothing overlaps the collective in either form (it needs the last gradient and feeds the optimizer), so all of
                                      # it is exposed; what the one-graph form removes is the two host enqueue boundaries around it
                                      "exposed_fraction_of_step": round(ar_us / (1e6 * el / args.steps), 4),
                                      "captured_in_graph": in_graph, "capture_probe": coll_why or ("ok" if in_graph else "not run"),
                                      "placement": ("inside the step's single hipGraph (RCCL all-reduce recorded on the capturing stream)" if in_graph else
                                                    "eager all-reduce (torch.distributed backend " + torch.distributed.get_backend() +
                                                    "; nccl = RCCL) between the two captured graphs of the step" if use_graph else "eager")}

    if world == 1 and use_graph and resident and DTYPE == "f32" and args.bwd_precision == "fp32" and not args.no_secondary:
        # secondary measurement, same run, same model and optimizer state: the step with the opt-in "bf16x3" backward products
        # (ops.set_bwd_precision; gradients stay inside the 1e-3 bar, tests/test_gpu_bf16.py).  NOT the headline: `value` above is exact fp32.
        graphed.close()
        ops.set_bwd_precision("bf16x3")
        try:
            g2 = GraphedTrainStep(model, opt, *batch(0), world=world, chains=args.chains, series=series, n_his=N_HIS, n_pred=N_PRED, rank=rank)
            for _ in range(args.warmup):
                g2()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                l2 = g2()
            torch.cuda.synchronize()
            el2 = time.perf_counter() - t1
            out["config"]["secondary_bwd_bf16x3"] = {"value": round(B_LOCAL * args.steps / el2, 2), "unit": "windows/s", "ms_per_step": round(1e3 * el2 / args.steps, 4),
                                                     "final_loss": round(float(l2.item()), 5),
                                                     "what": "same step with the backward kernels' matrix products as split-bf16 operands (three bf16 MFMAs per product, "
                                                             "~2^-16 relative; measured gradient error 1e-5 of max against the fp64 oracle, bar 1e-3); forward exact fp32"}
            g2.close()
        except Exception as e:  # noqa: BLE001
            out["config"]["secondary_bwd_bf16x3"] = {"error": repr(e)}
        ops.set_bwd_precision("fp32")
        graphed = None
        # second secondary: every matrix product of the fp32 blocks on v_mfma_f32_16x16x4_f32 (STGCN_MFMA_X6=0) instead of the default
        # "bf16x6" form of tc1_fwd / tc2_ln_fwd / tc1_bwd (fp32-accurate products on the bf16 matrix pipe)
        os.environ["STGCN_MFMA_X6"] = "0"
        try:
            g3 = GraphedTrainStep(model, opt, *batch(0), world=world, chains=args.chains, series=series, n_his=N_HIS, n_pred=N_PRED, rank=rank)
            for _ in range(args.warmup):
                g3()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                l3 = g3()
            torch.cuda.synchronize()
            el3 = time.perf_counter() - t1
            out["config"]["secondary_fp32_mfma"] = {"value": round(B_LOCAL * args.steps / el3, 2), "unit": "windows/s", "ms_per_step": round(1e3 * el3 / args.steps, 4),
                                                    "final_loss": round(float(l3.item()), 5),
                                                    "what": "same step with every matrix product on v_mfma_f32_16x16x4_f32 (STGCN_MFMA_X6=0: the form of rounds 1-5)"}
            g3.close()
        except Exception as e:  # noqa: BLE001
            out["config"]["secondary_fp32_mfma"] = {"error": repr(e)}
        finally:
            os.environ.pop("STGCN_MFMA_X6", None)
    if rank == 0 and not args.no_profile:
        # per-kernel durations with hipEvents on the launch stream, over the same K steps (second pass)
        # (eager launches: hipEvents cannot be recorded inside a graph replay; the kernels and shapes are the same)
        if graphed is not None:
            graphed.close()
        DropoutStream.disable_device_counter()
        L.dll.stgcn_profile_enable(1)
        ksteps = min(args.steps, 50 if N <= 1024 else 3)
        for _ in range(ksteps):
            train_step(model, opt, *batch(step_i), None)
            step_i += 1
        torch.cuda.synchronize()
        buf = C.create_string_buffer(1 << 15)
        L.check(L.dll.stgcn_profile_collect(buf, len(buf)), "stgcn_profile_collect")
        L.dll.stgcn_profile_enable(0)
        prof = json.loads(buf.value.decode())
        flops = stblock_flops_by_label(B_LOCAL, N, KS)
        per_step = {k: v["total_ms"] / ksteps for k, v in prof.items()}
        stblock_total = sum(v for k, v in flops.items() if k.startswith("_total"))
        mfma_kernels = {k: per_step[k] for k in flops if k in per_step and flops[k] > 0}
        dom = max(mfma_kernels, key=mfma_kernels.get)          # the kernel label (@ block) that costs most per step
        calls_per_step = prof[dom]["calls"] / ksteps
        dur_ms = prof[dom]["total_ms"] / prof[dom]["calls"]
        bf_mm = DTYPE == "bf16" or (dom.startswith("gso_gemm") and args.gc_precision != "fp32")
        peak = PEAK_BF16_MFMA_TFLOPS if bf_mm else PEAK_FP32_MFMA_TFLOPS
        ach = flops[dom] / (dur_ms * 1e-3) / 1e12
        tot_ms = sum(v for k, v in per_step.items() if not k.startswith(("head.", "adamw")))
        traffic, traffic_src = pmc_traffic(args.config, DTYPE)
        if args.gc_precision != "fp32" or args.bwd_precision != "fp32":
            traffic, traffic_src = {}, None   # (the committed counters are those of the default modes)
        st_labels = [k for k in per_step if not k.startswith(("head.", "adamw", "prepack", "mse", "reduce"))]
        st_traffic = sum(traffic.get(k, 0) * prof[k]["calls"] / ksteps for k in st_labels) if traffic else None
        nbytes = stblock_bytes_by_label(B_LOCAL, N, KS, 2 if DTYPE == "bf16" else 4) if N <= 512 else {}
        out["roofline"] = {"bound": "mfma", "timing_source": "hipEvent pairs around every launch, second (eager) pass over the same steps; the timed "
                                                             "value above is the hipGraph replay", "kernel": dom, "achieved": round(ach, 3), "peak": peak, "unit": "TFLOP/s",
                           "frac": round(ach / peak, 4), "traffic": traffic.get(dom), "traffic_unit": "bytes/launch",
                           "traffic_source": traffic_src,
                           "avg_launch_us": round(dur_ms * 1e3, 2), "launches_per_step": calls_per_step, "flops_per_launch": int(flops[dom]),
                           "stblock_kernels_ms_per_step": round(tot_ms, 4), "all_kernels_ms_per_step": round(sum(per_step.values()), 4),
                           "stblock_fwd_bwd_frac": round(stblock_total / (tot_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                           "stblock_flops_per_step": int(stblock_total),
                           "stblock_traffic_bytes": None if st_traffic is None else int(st_traffic),
                           "stblock_compulsory_bytes": cfg["compulsory_bytes"]["f32"],
                           "stblock_launches_per_step": int(round(sum(v["calls"] for k, v in prof.items()
                                                                        if not k.startswith(("head.", "adamw", "prepack", "mse", "reduce"))) / ksteps)),
                           "per_kernel_us_per_step": {k: round(v * 1e3, 2) for k, v in sorted(per_step.items())}}
        if dom in nbytes:      # the same launch against the HBM roofline; the bound is whichever limit is the slower one for this kernel
            gbs = nbytes[dom] / (dur_ms * 1e-3) / 1e9
            rl = out["roofline"]
            rl["algorithmic_bytes_per_launch"] = int(nbytes[dom])
            rl["hbm_gbs"], rl["hbm_frac"], rl["mfma_frac"] = round(gbs, 1), round(gbs / PEAK_HBM_GBS, 4), rl["frac"]
            if nbytes[dom] / (PEAK_HBM_GBS * 1e9) > flops[dom] / (peak * 1e12):
                rl.update(bound="hbm", achieved=round(gbs, 1), peak=PEAK_HBM_GBS, unit="GB/s", frac=round(gbs / PEAK_HBM_GBS, 4))
        if DTYPE == "bf16":
            rl = out["roofline"]
            comp = cfg["compulsory_bytes"]["bf16"]
            rl["stblock_compulsory_bytes"] = comp
            rl["stblock_fwd_bwd_frac"] = round(stblock_total / (tot_ms * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4)
            rl["stblock_hbm_frac"] = round(comp / (tot_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)
    if rank == 0 and world == 1 and not args.no_gpu_baseline:
        out["gpu_baseline"] = gpu_baseline(model, gso_t, cfg, B_LOCAL, N, dev)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        if N <= 1024:
            out["cpu_baseline"] = cpu_baseline(gso_np, cfg, B_LOCAL, args.cpu_budget)
            out["cpu_baseline"]["reference_published"] = {"value": {"c2": 159.2, "c3": 79.5}.get(args.config), "unit": "windows/s",
                                                          "source": "BASELINE.md section 2: the unmodified reference loop on 8 vCPU (survey container), fp32"}
        else:   # 8192 nodes: one oracle step costs ~1 TFLOP of fp32 operator products on the host -- a bounded sample at bs 2, no thread probing
            out["cpu_baseline"] = cpu_baseline(gso_np, cfg, 2, args.cpu_budget, probe=False)
    if rank == 0 and world == 1 and args.config == "c2" and not args.no_side_configs and not B_OVERRIDE:
        # the other single-GPU configurations of BASELINE.json, witnessed by the same invocation
        del model, opt
        torch.cuda.empty_cache()
        out["side_configs"] = {"c3_bf16": side_config("c3", args.steps, args.warmup), "c5_bf16": side_config("c5", min(args.steps, 30), min(args.warmup, 5))}
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
