"""Data-parallel training step of the reference loop body (main.py:164-171) on MI355X.

One process per GPU (torchrun / torch.distributed, backend "nccl" = RCCL over xGMI).  The model is
replicated (245 K parameters + a 171 KB operator), the minibatch is sharded: rank r takes windows
[step*B_global + r*B_local, ... + B_local) of the unshuffled index (the reference iterates with
shuffle=False, main.py:127), so W ranks x bs 32 reproduces one device at bs 32*W.  The only exchange is
ONE all-reduce per step over a flat fp32 gradient buffer (~0.98 MB for METR-LA): latency-bound, not
link-bound on xGMI, so no bucketing.

Parameters that never receive a gradient (10 unused align convs) are excluded from the flat buffer and
from the optimizer's work exactly like torch.optim.AdamW skips ``grad is None`` in the reference.
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def dist_env():
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    return rank, local_rank, world


def init_distributed(backend: Optional[str] = None):
    """One process per GPU.  The device is bound BEFORE the process group is created so that RCCL builds its
    communicator on the right GPU (and the first collective does not have to guess)."""
    rank, local_rank, world = dist_env()
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this driver stack
        if backend == "nccl":
            try:
                dist.init_process_group(backend=backend, rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
            except TypeError:      # older signature without device_id
                dist.init_process_group(backend=backend, rank=rank, world_size=world)
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    if world > 1:      # every data-parallel rank draws its own dropout masks (the shards are different windows of one global batch)
        from .layers import DropoutStream
        base = DropoutStream.seed - DropoutStream.rank_offset
        DropoutStream.rank_offset = rank        # survives a later DropoutStream.manual_seed(s) of user code (set-seed-after-init order)
        DropoutStream.manual_seed(base)
    return rank, local_rank, world


def probe_collective_capture(timeout_s: float = 60.0) -> Tuple[bool, str]:
    """Can this machine record an RCCL all-reduce inside a hipGraph and replay it?  Every rank runs the SAME small experiment in a CHILD
    process (its own process group on MASTER_PORT + 17, same GPU, a 1 MB buffer: warm-up collective, capture, three replays, value check)
    under a watchdog: a capture or replay that hangs -- the failure mode that cannot be caught with try / except -- costs ``timeout_s``
    and one killed child, not the run.  The verdicts are combined over the ranks (all must succeed).  Returns (ok, reason)."""
    rank, local_rank, world = dist_env()
    if not (dist.is_available() and dist.is_initialized()) or world <= 1:
        return False, "no process group"
    if dist.get_backend() != "nccl":
        return False, f"backend {dist.get_backend()}: its collectives are host calls, not stream operations"
    ok, why = run_collective_capture_child(dict(os.environ), timeout_s)
    flag = torch.tensor([1 if ok else 0], device=torch.device("cuda", local_rank))
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) == 0 and ok:
        ok, why = False, "another rank's probe failed"
    return ok, why


def capture_child_env(env: dict) -> dict:
    """Environment of the probe's child: the parent's rank variables, the rendezvous port shifted by 17, and WITHOUT the launcher's
    TORCHELASTIC_* variables -- under ``torch.distributed.run`` TORCHELASTIC_USE_AGENT_STORE makes ``env://`` rendezvous attach to the
    agent's store as a client; on the shifted port nobody serves one, so every child would wait for its timeout and the probe would
    always say "hung".  Without them rank 0's child hosts its own store."""
    env = {k: v for k, v in env.items() if not k.startswith("TORCHELASTIC_")}
    env["MASTER_PORT"] = str(int(env.get("MASTER_PORT", "29500")) + 17)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def run_collective_capture_child(env: dict, timeout_s: float = 60.0) -> Tuple[bool, str]:
    """One rank's share of ``probe_collective_capture``: the experiment in a child process (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from
    ``env``, port shifted by 17 so that it cannot meet the parent's group), killed after ``timeout_s``."""
    import subprocess
    import sys
    code = (
        "import os, torch, torch.distributed as dist\n"
        "r, lr, w = int(os.environ['RANK']), int(os.environ['LOCAL_RANK']), int(os.environ['WORLD_SIZE'])\n"
        "torch.cuda.set_device(lr)\n"
        "dist.init_process_group('nccl', rank=r, world_size=w, device_id=torch.device('cuda', lr))\n"
        "t = torch.full((1 << 18,), float(r + 1), device='cuda')\n"
        "dist.all_reduce(t); torch.cuda.synchronize()\n"
        "t.fill_(float(r + 1))\n"
        "s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())\n"
        "g = torch.cuda.CUDAGraph()\n"
        "with torch.cuda.graph(g, capture_error_mode='thread_local'):\n"
        "    dist.all_reduce(t)\n"
        "    t.mul_(1.0 / w)\n"
        "for _ in range(3):\n"
        "    t.fill_(float(r + 1)); g.replay()\n"
        "torch.cuda.synchronize()\n"
        "want = sum(range(1, w + 1)) / w\n"
        "assert abs(float(t[0]) - want) < 1e-6 and abs(float(t[-1]) - want) < 1e-6, (float(t[0]), want)\n"
        "dist.destroy_process_group()\n"
        "print('CAPTURE_OK')\n")
    env = capture_child_env(env)
    ok, why = False, ""
    try:
        p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=timeout_s)
        ok = p.returncode == 0 and "CAPTURE_OK" in p.stdout
        why = "" if ok else ("probe failed: " + (p.stderr.strip().splitlines() or ["rc %d" % p.returncode])[-1][:200])
    except subprocess.TimeoutExpired:
        why = f"probe hung for {timeout_s:.0f} s (killed)"
    except Exception as e:  # noqa: BLE001
        why = "probe could not run: " + repr(e)[:200]
    return ok, why


def sync_operators(model, src: int = 0) -> None:
    """Data-parallel replicas must train with ONE graph shift operator: ``gso`` is a plain attribute (not a parameter, not a
    buffer, never all-reduced -- layers.py:128/179 of the reference), and the reference's ``calc_chebynet_gso`` depends on numpy's
    global RNG state (script/utility.py:59-76), so differently seeded ranks would silently hold different operators.  Broadcasts
    every fused block's ``gso`` from rank ``src`` (no-op without a process group)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() <= 1:
        return
    from .layers import STConvBlock
    seen = {}
    for m in model.modules():
        if isinstance(m, STConvBlock) and torch.is_tensor(m.gso):
            g = seen.get(id(m.gso))
            if g is None:
                # the collective runs on the process group's device (RCCL needs a tensor on this rank's GPU; `gso` is a plain attribute
                # that .to(device) never moved, so it is often still a CPU tensor) and the result goes back where the operator lived
                home = m.gso.device
                comm_dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
                t = m.gso.detach().to(comm_dev, copy=True).contiguous()
                dist.broadcast(t, src=src)
                g = t.to(home)
                seen[id(m.gso)] = g
            m.gso = g
            m.graph_conv.gso = g
            gc = getattr(m.graph_conv, "cheb_graph_conv", None) or getattr(m.graph_conv, "graph_conv", None)
            if gc is not None:
                gc.gso = g
            m._gso_cache = None


class FlatGradAllReduce:
    """Sum-all-reduce of every live gradient in one collective, then scale by 1/world."""

    def __init__(self, params: List[torch.nn.Parameter], world: int):
        self.params = list(params)
        self.world = world
        self.flat: Optional[torch.Tensor] = None
        self.live: Optional[List[torch.nn.Parameter]] = None

    def __call__(self):
        if self.world <= 1:
            return
        live = [p for p in self.params if p.grad is not None]
        if self.live is None or len(live) != len(self.live):
            self.live = live
            n = sum(p.numel() for p in live)
            self.flat = torch.empty(n, dtype=torch.float32, device=live[0].device)
        grads = [p.grad.reshape(-1) for p in live]
        torch.cat(grads, out=self.flat)
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        self.flat.mul_(1.0 / self.world)
        torch._foreach_copy_([p.grad.reshape(-1) for p in live], list(self.flat.split([p.numel() for p in live])))


def make_optimizer(model: torch.nn.Module, lr: float = 1e-3, weight_decay: float = 1e-3, name: str = "adamw",
                   capturable: bool = False):
    """main.py:147-154 (adamw is the default; nadamw is NAdam with decoupled decay).
    ``capturable``: keep the step counters on the device so that ``step()`` can live inside a hipGraph."""
    params = list(model.parameters())
    if name == "adamw":
        from .optim import AdamW        # one HIP kernel per step (stgcn_adamw_step)
        return AdamW(params, lr=lr, weight_decay=weight_decay, capturable=capturable)
    if name == "nadamw":
        return torch.optim.NAdam(params, lr=lr, weight_decay=weight_decay, decoupled_weight_decay=True)
    raise ValueError(f"ERROR: The {name} optimizer is undefined.")   # main.py:154


def fwd_loss_bwd(model, x, y):
    """forward -> MSELoss -> backward (main.py:166-168) with the loss formed inside the fused head's backward (``ops.mse_backward``;
    one loss launch + ``pred.backward(dpred)`` for a model without the fused head).  Returns the loss (shape [])."""
    from . import ops
    y_pred = model(x).reshape(len(x), -1)
    return ops.mse_backward(y_pred, y)[0]


def check_in_launch_waits(model) -> None:
    """Raise if a fused operator's last forward had an in-launch wait that gave up (the tiles of a LayerNorm slab / window wait for each
    other inside ONE launch; a wait is bounded, and a starved tile writes NaN and sets a sticky word instead of hanging the device).  The
    NaN reaches the loss anyway -- this names the operator and the slab / window.  Synchronises the stream: call it where the loss is read
    (once per epoch, after a timed loop), not inside the step."""
    from .layers import OutputBlock, STConvBlock
    for name, m in model.named_modules():
        if isinstance(m, (STConvBlock, OutputBlock)):
            w = m.chain_status()
            if w:
                what = "(b, t) slab" if isinstance(m, STConvBlock) else "window"
                raise RuntimeError(f"{name or type(m).__name__}: an in-launch wait of the last forward gave up on {what} {w - 1} -- its outputs are NaN "
                                   f"(the device was not this launch's alone for stgcn_set_chain_spin_ticks; the next weight pack re-arms the words)")


def fused_tail_supported(model, live_params) -> bool:
    """The fused step tail (``GradSink`` / ``stgcn_grad_flush``) writes the gradient of every parameter OWNED BY A FUSED OPERATOR
    (``STConvBlock``, a supported ``OutputBlock``) and nothing else.  A live parameter outside of them (the two ``nn.Linear`` of the
    Ko == 0 head, models.py:46-51, or any user module around the model) gets its gradient from autograd, which would ACCUMULATE
    into the never-zeroed arena view installed as ``.grad``: such models take the plain step (zero_grad / backward / step)."""
    from . import ops
    from .layers import OutputBlock, STConvBlock
    owned = set()
    for m in model.modules():
        if isinstance(m, STConvBlock) or (isinstance(m, OutputBlock) and ops.head_supported(m.cfg)):
            owned.update(id(p) for p in m.parameters())
    return all(id(p) in owned for p in live_params)


class GradArena:
    """One flat fp32 buffer holding the gradient of every live parameter (those the model actually uses: the reference
    leaves ``.grad`` None for the 10 idle align convs) in ``model.parameters()`` order, with one view
    per parameter installed as ``param.grad``.  It is at once the destination of ``ops.GradSink.flush`` (no per-step
    gradient allocations, no autograd accumulation) and the data-parallel all-reduce buffer (no flatten / unflatten
    copies around the collective)."""

    def __init__(self, live_params: List[torch.nn.Parameter]):
        from . import ops
        self.params = list(live_params)
        dev = self.params[0].device
        self.flat = torch.zeros(sum(p.numel() for p in self.params), dtype=torch.float32, device=dev)
        self.grads = {}
        off = 0
        for p in self.params:
            self.grads[p] = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()
        self.sink = ops.GradSink(self.grads)

    def install(self):
        for p, g in self.grads.items():
            p.grad = g


def fused_train_step(model, optimizer, x, y, arena: GradArena, world: int = 1, all_reduce=None):
    """The loop body of main.py:165-169 with the step tail fused: forward -> backward with the MSE loss formed inside the head and
    deferred reductions -> ONE launch for all gradient reductions + AdamW (world == 1), or reductions into the flat arena ->
    ``all_reduce(arena.flat)`` -> AdamW (world > 1; the 1/world of the gradient mean rides on the loss gradient).
    ``param.grad`` are the arena views (overwritten every step, never accumulated)."""
    from . import ops
    if not fused_tail_supported(model, arena.params):
        raise RuntimeError("fused_train_step: the model has live parameters outside the fused operators (see fused_tail_supported); "
                           "use train_step")
    arena.install()
    with ops.grad_sink_scope(arena.sink):
        y_pred = model(x).reshape(len(x), -1)
        loss = ops.mse_backward(y_pred, y, grad_scale=1.0 / world)      # (valid after the flush below)
    if world <= 1 and hasattr(optimizer, "flush_with"):
        optimizer.flush_with(arena.sink, arena.grads)
    else:
        arena.sink.flush(stream=torch.cuda.current_stream(x.device).cuda_stream if x.is_cuda else None)
        if world > 1 and all_reduce is not None:
            all_reduce(arena.flat)
        optimizer.step()
    return loss[0]


def train_step(model, optimizer, x, y, allreduce: Optional[FlatGradAllReduce] = None):
    """zero_grad -> forward -> MSELoss -> backward -> [all-reduce] -> optimizer.step (main.py:165-169).
    Returns the loss tensor (no host sync: the reference's per-step .item() at main.py:170 is deferred)."""
    from .layers import DropoutStream
    optimizer.zero_grad(set_to_none=True)
    loss = fwd_loss_bwd(model, x, y)
    if allreduce is not None:
        allreduce()
    optimizer.step()
    DropoutStream.advance()      # (device-counter mode only: eager mode consumes a fresh host offset per forward)
    return loss.detach()


def tail_step(model, optimizer, x, y, world: int = 1, rank: int = 0):
    """The partial last minibatch of an epoch (the reference iterates with ``DataLoader(drop_last=False)``, main.py:126-131, so its
    last step sees n < batch_size windows).  ``x`` / ``y`` are the GLOBAL tail batch (n windows, the same tensors on every rank);
    rank r takes a contiguous share of ceil(n / world) windows -- possibly none -- and weights its loss gradient by n_local / n, so
    the SUM all-reduce of the gradients is exactly the gradient of the mean loss over the n windows (``weight by local count``,
    SURVEY.md section 8e).  Eager launches: a captured step has a fixed batch shape.  Returns the global mean loss (tensor)."""
    from . import ops
    from .layers import DropoutStream
    n = len(x)
    per = (n + world - 1) // world
    lo, hi = min(rank * per, n), min(rank * per + per, n)
    optimizer.zero_grad(set_to_none=True)
    dev = x.device
    loss_sum = torch.zeros(1, dtype=torch.float32, device=dev)
    # A live GraphedTrainStep may have put its step counters (dropout position, AdamW step count, series window index) on the model's
    # weight-pack launch (``model._step_counters``): this eager step must not let that launch advance the WINDOW index (the tail batch
    # is handed over explicitly; the next replay has to start at the window the index already points to), so the counters are suspended
    # around the forward and the two that a tail step does consume -- dropout position and optimizer step count -- are advanced below.
    folded = getattr(model, "_step_counters", None)
    if folded is not None:
        model._step_counters = None
    try:
        if hi > lo:
            pred = model(x[lo:hi]).reshape(hi - lo, -1)
            loss = ops.mse_backward(pred, y[lo:hi].contiguous(), grad_scale=(hi - lo) / n)
            loss_sum = loss * ((hi - lo) / n)
    finally:
        if folded is not None:
            model._step_counters = folded
    if world > 1:
        params = [p for p in model.parameters() if p.requires_grad]
        live = torch.tensor([1.0 if p.grad is not None else 0.0 for p in params], device=dev)
        dist.all_reduce(live, op=dist.ReduceOp.MAX)          # a rank with an empty share has no gradients at all
        keep = [p for p, l in zip(params, live.tolist()) if l > 0]
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in keep])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        for p, g in zip(keep, flat.split([p.numel() for p in keep])):
            p.grad = g.view_as(p).clone()
        dist.all_reduce(loss_sum, op=dist.ReduceOp.SUM)
    if getattr(optimizer, "trainer_owns_step", False) and hasattr(optimizer, "device_step_counter"):
        optimizer.device_step_counter(dev).add_(1)       # (the suspended pack launch would have counted this step)
    optimizer.step()
    DropoutStream.advance()      # device-counter mode: the tail batch and the next replay must not share dropout masks
    return loss_sum[0].detach()


def chained_fwd_bwd(model, x, y, chains: int, streams=None):
    """zero_grad'ed forward + MSE + backward of one minibatch as ``chains`` independent micro-batch chains.

    Every kernel of this path is a 10-40 us launch whose workgroups walk load -> MFMA -> store together, so a single
    in-order chain leaves the memory system idle while the MFMA pipes run and vice versa.  The windows of a minibatch are
    independent (LayerNorm is per (b, t) slab), so the minibatch is cut into ``chains`` slices whose
    complete forward + backward run CONCURRENTLY on separate HIP streams (one fork at the start, one join at the end;
    per-chain workspaces through ``ops.chain_scope``); their parameter gradients are summed in a fixed order after the
    join, which equals the gradient of the mean loss over the whole minibatch.  ``streams[c]`` (c >= 1) are the side
    streams; ``None`` runs the chains back to back on the current stream (CPU emulator / tests).
    Sets ``p.grad`` and returns the detached mean loss."""
    from contextlib import nullcontext
    from . import ops
    params = [p for p in model.parameters() if p.requires_grad]
    B = len(x)
    assert chains >= 1 and B % chains == 0, f"batch {B} does not split into {chains} chains"
    Bc = B // chains
    use_streams = streams is not None and x.is_cuda and chains > 1
    main = torch.cuda.current_stream(x.device) if use_streams else None
    losses, grads = [], []
    for c in range(chains):
        s = streams[c] if (use_streams and c > 0) else None
        if s is not None:
            s.wait_stream(main)                       # fork: inputs / parameters are ready on the main stream
        with (torch.cuda.stream(s) if s is not None else nullcontext()), ops.chain_scope(c):
            xc, yc = x[c * Bc:(c + 1) * Bc], y[c * Bc:(c + 1) * Bc]
            pred = model(xc).reshape(Bc, -1)
            loss = torch.nn.functional.mse_loss(pred, yc) * (1.0 / chains)
            grads.append(torch.autograd.grad(loss, params, allow_unused=True))
            losses.append(loss.detach())
    if use_streams:
        for c in range(1, chains):
            main.wait_stream(streams[c])              # join
    live = [i for i, g in enumerate(grads[0]) if g is not None]
    acc = [grads[0][i] for i in live]
    for c in range(1, chains):
        torch._foreach_add_(acc, [grads[c][i] for i in live])
    for i, g in zip(live, acc):
        params[i].grad = g
    total = losses[0]
    for l in losses[1:]:
        total = total + l
    return total


class GraphedTrainStep:
    """The loop body of main.py:165-169 captured once into hipGraph(s) and replayed.

    The eager step is host-bound on MI355X (the GPU needs ~1 ms of kernels, Python + ~100 launches need
    more), so the whole step -- forward, MSE, backward, AdamW, dropout-counter bump -- is recorded with
    ``torch.cuda.graph`` and replayed with one host call.  Kernel arguments are frozen by capture, hence:
      * inputs are copied into static buffers before each replay;
      * the dropout offset comes from a device counter (``DropoutStream.use_device_counter``);
      * the optimizer is built ``capturable`` (device-side step counts).
    With world > 1 the step is split into two graphs around ONE eager RCCL all-reduce of the flat gradient
    buffer (fwd+bwd+reductions into the flat ``GradArena`` | all-reduce | AdamW).
    ``fused`` (default on, ``STGCN_FUSED_STEP=0`` disables): the step tail of ``fused_train_step``.
    """

    def __init__(self, model, optimizer, x_example: torch.Tensor, y_example: torch.Tensor, world: int = 1, warmup: int = 3,
                 chains: int = 1, fused: Optional[bool] = None, series: Optional[torch.Tensor] = None, n_his: int = 12,
                 n_pred: int = 3, rank: int = 0, capture_collective: bool = False):
        """``series``: optional resident (time, N) float32 device tensor (already z-scored).  The step then takes its windows
        straight from it (device-side windowing): window b of the minibatch is rows
        [s + b, s + b + n_his) of the series (read in place through a strided view, no (num, 1, n_his, N) tensor, no per-step
        input copies), its label row s + b + n_his + n_pred - 1 (script/dataloader.py:32-47), and s advances by the global
        batch on the device every replay (unshuffled order like main.py:127, wrapping at the end of the series).  Call the
        step without arguments; ``x_example`` / ``y_example`` only give the batch size.

        ``capture_collective`` (world > 1): record the gradient all-reduce INSIDE the graph -- one graph per step (forward, backward,
        reductions, RCCL all-reduce, AdamW), no host round trip around the collective.  Only for a backend whose collectives are stream
        operations (nccl = RCCL); the caller is expected to have probed that such a capture works on this machine
        (``probe_collective_capture``: a capture that hangs instead of raising would otherwise take the run with it)."""
        from .layers import DropoutStream
        assert x_example.is_cuda, "hipGraph capture needs the MI355X path"
        self.model, self.opt, self.world = model, optimizer, world
        dev = x_example.device
        self.series, self.index, self._index_bump = series, None, None
        self.chains = int(chains)                 # micro-batch chains on concurrent streams (chained_fwd_bwd)
        if fused is None:
            fused = os.environ.get("STGCN_FUSED_STEP", "1") != "0"
        # fused step tail (GradArena / GradSink): one-launch loss, one launch for all gradient reductions + AdamW
        # (decided for good after the first plain step has shown which parameters are live: fused_tail_supported)
        self.fused = bool(fused) and self.chains == 1 and hasattr(optimizer, "flush_with")
        self.arena: Optional[GradArena] = None
        self.streams = [None] + [torch.cuda.Stream(device=dev) for _ in range(self.chains - 1)]
        if series is None:
            self.x = torch.empty_like(x_example)
            self.y = torch.empty_like(y_example)
            self.x.copy_(x_example)
            self.y.copy_(y_example)
        else:
            from . import ops
            B, N = len(x_example), series.shape[1]
            assert series.is_cuda and series.dtype == torch.float32 and series.is_contiguous() and B > 1
            num = series.shape[0] - n_his - n_pred                      # windows as the reference counts them (dataloader.py:36)
            # whole global minibatches only: the reference's DataLoader(drop_last=False) also yields the partial tail batch
            # (main.py:126-131); a captured step has a fixed batch shape, so the tail (< B * world windows per epoch) is left to
            # the eager path (``train.tail_step``).  The constructor itself runs `warmup` + 1 REAL optimizer steps (capture needs
            # warmed-up allocator state): the window position after construction is (warmup + 1) * B * world.
            usable = num // (B * world) * (B * world)
            assert usable > 0, "series shorter than one global minibatch of windows"
            # a model with bf16 activations reads its windows from a bf16 copy of the series (made once; the labels stay fp32)
            cd = getattr(model, "compute_dtype", None)
            self.series_x = series if cd in (None, series.dtype) else series.to(cd)
            self.x = torch.as_strided(self.series_x, (B, 1, n_his, N), (N, n_his * N, N, 1))   # window b = rows [b, b + n_his)
            self.y = series[n_his + n_pred - 1:n_his + n_pred - 1 + B]                         # label rows, (B, N) contiguous
            self.index = torch.full((1,), rank * B, dtype=torch.int64, device=dev)             # first window of this rank
            self._index_bump = (self.index, B * world, usable)
            ops.bind_input_index(self.x, self.index, N)
            ops.bind_input_index(self.y, self.index, N)
        if DropoutStream.counter is None or DropoutStream.counter.device != dev:
            DropoutStream.use_device_counter(dev)
        self.counter = DropoutStream.counter     # the captured kernels hold this address: keep it alive with the graph
        self.loss = None
        self.flat = None
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        self.fold = False       # step counters advanced by the weight-pack launch instead of two tiny launches of their own
        with torch.cuda.stream(side):
            for i in range(max(warmup, 2 if self.fused else 1)):
                if self.fused and i > 0 and self.arena is None:
                    live = [p for p in model.parameters() if p.grad is not None]      # the first (plain) step showed them
                    if not fused_tail_supported(model, live):
                        self.fused = False
                if self.fused and i > 0:
                    if self.arena is None:
                        self.arena = GradArena([p for p in model.parameters() if p.grad is not None])
                        # the step counters ride on the weight-pack launch for every world size: with world > 1 the second graph is then
                        # exactly ONE launch (AdamW over the all-reduced arena), no add_ / remainder_ launches of the counters
                        self._try_fold_counters(dev)         # (is itself one fused training step)
                        continue
                    self._eager_fused_once()
                else:
                    self._eager_once()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        if not self.fused:
            self.opt.zero_grad(set_to_none=True)
        self.capture_collective = bool(capture_collective) and world > 1
        self.g1 = torch.cuda.CUDAGraph()
        # with a process group alive its watchdog thread polls events while this thread captures: "thread_local" keeps such calls of OTHER
        # threads from invalidating the capture (the default "global" mode is for single-threaded programs)
        cap = dict(capture_error_mode="thread_local") if world > 1 else {}
        with torch.cuda.graph(self.g1, **cap):
            if self.fused:
                self.loss = self._fused_fwd_bwd()
                if world > 1:
                    self.flat = self.arena.flat
                    if self.capture_collective:      # the whole step in ONE graph: collective and optimizer follow on the capturing stream
                        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
                        self.opt.step()
                        if not self.fold:
                            DropoutStream.advance()
                            self._advance_series()
                else:
                    self.opt.flush_with(self.arena.sink, self.arena.grads, bump_step=not self.fold)
                    if not self.fold:
                        DropoutStream.advance()
                        self._advance_series()
            else:
                self.loss = self._fwd_bwd()
                if world > 1:
                    self.live = [p for p in model.parameters() if p.grad is not None]
                    self.flat = torch.cat([p.grad.reshape(-1) for p in self.live])
                    if self.capture_collective:
                        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
                        self.flat.mul_(1.0 / world)
                        torch._foreach_copy_([p.grad.reshape(-1) for p in self.live], list(self.flat.split([p.numel() for p in self.live])))
                        self.opt.step()
                        DropoutStream.advance()
                        self._advance_series()
                else:
                    self.opt.step()
                    DropoutStream.advance()
                    self._advance_series()
        self.g2 = None
        if world > 1 and not self.capture_collective:
            self.g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g2, pool=self.g1.pool(), **cap):
                if not self.fused:
                    self.flat.mul_(1.0 / world)
                    torch._foreach_copy_([p.grad.reshape(-1) for p in self.live], list(self.flat.split([p.numel() for p in self.live])))
                self.opt.step()
                if not self.fold:
                    DropoutStream.advance()
                    self._advance_series()
        # one full replay inside the constructor: a capture / replay problem surfaces here (where the caller can
        # still fall back to eager launches of the same kernels), not inside a timed loop
        self(x_example, y_example)
        torch.cuda.synchronize(dev)

    def check(self) -> None:
        """``check_in_launch_waits`` on the step's model (synchronises: for the places where the loss is read)."""
        check_in_launch_waits(self.model)

    def close(self):
        """Undo the process-global state a captured step leaves behind (device dropout counter, step counters on the model's pack
        launch, input-index bindings) so that later EAGER steps on the same model behave like clean eager steps."""
        from . import ops
        from .layers import DropoutStream
        if getattr(self.model, "_step_counters", None) is not None:
            self.model._step_counters = None
        if self.series is not None:
            ops.unbind_input_index(self.x)
            ops.unbind_input_index(self.y)
        if DropoutStream.counter is self.counter:
            DropoutStream.disable_device_counter()
        if hasattr(self.opt, "trainer_owns_step"):
            self.opt.trainer_owns_step = False

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def _fwd_bwd(self):
        if self.chains > 1:
            return chained_fwd_bwd(self.model, self.x, self.y, self.chains, self.streams)
        return fwd_loss_bwd(self.model, self.x, self.y)

    def _fused_fwd_bwd(self):
        """forward, one-launch loss + gradient (carrying the 1/world of the gradient mean), backward with deferred reductions;
        with world > 1 the reductions are flushed into the flat arena here (the all-reduce follows outside the graph)."""
        from . import ops
        self.arena.install()
        with ops.grad_sink_scope(self.arena.sink):
            y_pred = self.model(self.x).reshape(len(self.x), -1)
            loss = ops.mse_backward(y_pred, self.y, grad_scale=1.0 / self.world)      # (written by the gradient flush of the step)
        if self.world > 1:
            self.arena.sink.flush(stream=torch.cuda.current_stream(self.x.device).cuda_stream)
        return loss[0]

    def _try_fold_counters(self, dev):
        """Let the model's weight-pack launch advance the dropout position and the optimizer step count; verified by one
        eager step (a model whose forward does not go through stgcn_prepack keeps the explicit bumps)."""
        from .layers import DropoutStream
        step_t = self.opt.device_step_counter(dev)
        self.model._step_counters = [(DropoutStream.counter, DropoutStream.SITE_STRIDE, 0), (step_t, 1, 0)]
        i0 = None
        if self._index_bump is not None:
            # unfolded steps advance the window position AFTER the step, the pack launch advances it BEFORE: step back once
            idx, inc, mod = self._index_bump
            i0 = int(idx.item())
            idx.sub_(inc).remainder_(mod)
            self.model._step_counters.append(self._index_bump)
        c0, s0 = int(DropoutStream.counter.item()), int(step_t.item())
        self.fold = True
        self.opt.trainer_owns_step = True       # the pack launch counts the steps: AdamW.step() must not count them again
        self._eager_fused_once()
        if int(DropoutStream.counter.item()) != c0 + DropoutStream.SITE_STRIDE or int(step_t.item()) != s0 + 1:
            self.model._step_counters = None          # the pack launch did not run: advance them the explicit way
            DropoutStream.counter.fill_(c0 + DropoutStream.SITE_STRIDE)
            step_t.fill_(s0 + 1)
            if i0 is not None:
                self._index_bump[0].fill_((i0 + self._index_bump[1]) % self._index_bump[2])
            self.fold = False
            self.opt.trainer_owns_step = False

    def _eager_fused_once(self):
        from .layers import DropoutStream
        self._fused_fwd_bwd()
        if self.world > 1:
            dist.all_reduce(self.arena.flat)
            self.opt.step()                  # (fold: the pack launch counted the step, trainer_owns_step keeps step() from counting again)
            if not self.fold:
                DropoutStream.advance()
                self._advance_series()
        else:
            self.opt.flush_with(self.arena.sink, self.arena.grads, bump_step=not self.fold)
            if not self.fold:
                DropoutStream.advance()
                self._advance_series()

    def _eager_once(self):
        from .layers import DropoutStream
        self.opt.zero_grad(set_to_none=True)
        self._fwd_bwd()
        if self.world > 1:
            live = [p for p in self.model.parameters() if p.grad is not None]
            flat = torch.cat([p.grad.reshape(-1) for p in live])
            dist.all_reduce(flat)
            flat.mul_(1.0 / self.world)
            torch._foreach_copy_([p.grad.reshape(-1) for p in live], list(flat.split([p.numel() for p in live])))
        self.opt.step()
        DropoutStream.advance()
        self._advance_series()

    def _advance_series(self):
        """the window position when no pack launch carries it (world > 1 or an unfolded step): two tiny launches"""
        if self._index_bump is not None and not self.fold:
            idx, inc, mod = self._index_bump
            idx.add_(inc).remainder_(mod)

    def __call__(self, x: Optional[torch.Tensor] = None, y: Optional[torch.Tensor] = None) -> torch.Tensor:
        if self.series is None:
            self.x.copy_(x, non_blocking=True)
            self.y.copy_(y, non_blocking=True)
        self.g1.replay()
        if self.g2 is not None:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            self.g2.replay()
        return self.loss
