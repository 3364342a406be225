"""CPU oracle for the STGCN ST-Conv-block training path  --  TEST INFRASTRUCTURE ONLY.

This file is a functional (stateless) restatement of the arithmetic of the
reference's hot path.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it, and only as the checker.
The product path (``stgcn_amd``) never imports anything from ``oracle/``.

Parity status: the reference ships no tests / golden vectors,
so the oracle is pinned against *outputs of the reference itself*, produced in
the build container by ``tests/golden/make_golden.py`` (which imports the
Python reference from /root/reference) and committed as ``tests/golden/*.npz``.
``tests/test_oracle_golden.py`` replays those fixtures through this file.

Every function cites the reference lines it follows (paths relative to the
reference checkout).  Parameters are passed as a flat dict keyed by the
reference's ``state_dict`` names, so the oracle also pins the
checkpoint-key contract.

Arithmetic lives in ATen (torch CPU) exactly like the reference's does
(model/layers.py delegates everything to conv2d / einsum / layer_norm); the
oracle therefore runs in float32 or float64 depending on the dtype of the
tensors it is handed.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------- config
@dataclass
class OracleConfig:
    """Hyper-parameters the reference reads from ``args`` (model/models.py:32-42)."""
    Kt: int = 3
    Ks: int = 3
    n_his: int = 12
    act_func: str = "glu"
    graph_conv_type: str = "cheb_graph_conv"
    enable_bias: bool = True
    droprate: float = 0.5
    blocks: List[List[int]] = field(default_factory=lambda: [[1], [64, 16, 64], [64, 16, 64], [128, 128], [1]])

    @property
    def n_st_blocks(self) -> int:
        return len(self.blocks) - 3          # model/models.py:31

    @property
    def Ko(self) -> int:
        return self.n_his - self.n_st_blocks * 2 * (self.Kt - 1)   # model/models.py:34


def default_blocks(n_his: int, Kt: int, stblock_num: int) -> List[List[int]]:
    """Channel plan built by main.py:80-92."""
    Ko = n_his - (Kt - 1) * 2 * stblock_num
    blocks: List[List[int]] = [[1]]
    for _ in range(stblock_num):
        blocks.append([64, 16, 64])
    if Ko == 0:
        blocks.append([128])
    elif Ko > 0:
        blocks.append([128, 128])
    blocks.append([1])
    return blocks


# --------------------------------------------------------------------------- layers
def align(x: Tensor, c_in: int, c_out: int, w: Optional[Tensor], b: Optional[Tensor]) -> Tensor:
    """model/layers.py:14-23 -- 1x1 conv when shrinking, zero-pad channels when growing."""
    if c_in > c_out:
        return F.conv2d(x, w, b)                                   # layers.py:15-16
    if c_in < c_out:
        B, _, T, N = x.shape
        pad = torch.zeros(B, c_out - c_in, T, N, dtype=x.dtype, device=x.device)
        return torch.cat([x, pad], dim=1)                          # layers.py:17-19
    return x                                                        # layers.py:20-21


def temporal_conv(x: Tensor, p: Dict[str, Tensor], prefix: str, Kt: int, c_in: int, c_out: int,
                  act_func: str) -> Tensor:
    """model/layers.py:87-120 -- gated causal (Kt x 1) convolution.

    ``x`` is logical (B, c_in, T, N); result is (B, c_out, T-Kt+1, N).
    """
    x_in = align(x, c_in, c_out, p.get(prefix + "align.align_conv.weight"),
                 p.get(prefix + "align.align_conv.bias"))[:, :, Kt - 1:, :]        # layers.py:88
    z = F.conv2d(x, p[prefix + "causal_conv.weight"], p.get(prefix + "causal_conv.bias"))  # :89 (valid conv, :55)
    if act_func in ("glu", "gtu"):
        x_p = z[:, :c_out]                                          # layers.py:92
        x_q = z[:, -c_out:]                                         # layers.py:93
        if act_func == "glu":
            return (x_p + x_in) * torch.sigmoid(x_q)                # layers.py:105
        return torch.tanh(x_p + x_in) * torch.sigmoid(x_q)          # layers.py:109
    if act_func == "relu":
        return torch.relu(z + x_in)                                 # layers.py:112
    if act_func == "silu":
        return F.silu(z + x_in)                                     # layers.py:115
    raise NotImplementedError(f"ERROR: The activation function {act_func} is not implemented.")  # :118


def cheb_graph_conv(x: Tensor, gso: Tensor, weight: Tensor, bias: Optional[Tensor]) -> Tensor:
    """model/layers.py:143-172.  x logical (B, c, T, N) -> (B, T, N, c_out)."""
    Ks = weight.shape[0]
    x = x.permute(0, 2, 3, 1)                                       # layers.py:145
    if Ks - 1 < 0:
        raise ValueError(f"ERROR: the graph convolution kernel size Ks has to be a positive integer, "
                         f"but received {Ks}.")                     # layers.py:147-148
    x_list = [x]
    if Ks - 1 >= 1:
        x_list.append(torch.einsum("hi,btij->bthj", gso, x))        # layers.py:154/158
    for k in range(2, Ks):
        x_list.append(torch.einsum("hi,btij->bthj", 2 * gso, x_list[k - 1]) - x_list[k - 2])  # :161
    xs = torch.stack(x_list, dim=2)                                 # layers.py:163
    out = torch.einsum("btkhi,kij->bthj", xs, weight)               # layers.py:165
    if bias is not None:
        out = out + bias                                            # layers.py:168
    return out


def kipf_graph_conv(x: Tensor, gso: Tensor, weight: Tensor, bias: Optional[Tensor]) -> Tensor:
    """model/layers.py:194-206."""
    x = x.permute(0, 2, 3, 1)                                       # layers.py:196
    first = torch.einsum("hi,btij->bthj", gso, x)                   # layers.py:198
    second = torch.einsum("bthi,ij->bthj", first, weight)           # layers.py:199
    if bias is not None:
        second = second + bias                                      # layers.py:202
    return second


def graph_conv_layer(x: Tensor, gso: Tensor, p: Dict[str, Tensor], prefix: str, graph_conv_type: str,
                     c_in: int, c_out: int) -> Tensor:
    """model/layers.py:222-231 -- align, graph conv, permute back, residual."""
    x_gc_in = align(x, c_in, c_out, p.get(prefix + "align.align_conv.weight"),
                    p.get(prefix + "align.align_conv.bias"))        # layers.py:223
    if graph_conv_type == "cheb_graph_conv":
        x_gc = cheb_graph_conv(x_gc_in, gso, p[prefix + "cheb_graph_conv.weight"],
                               p.get(prefix + "cheb_graph_conv.bias"))      # layers.py:225
    elif graph_conv_type == "graph_conv":
        x_gc = kipf_graph_conv(x_gc_in, gso, p[prefix + "graph_conv.weight"],
                               p.get(prefix + "graph_conv.bias"))           # layers.py:227
    else:
        raise ValueError(f"unknown graph_conv_type {graph_conv_type}")
    return x_gc.permute(0, 3, 1, 2) + x_gc_in                       # layers.py:228-229


def dropout(x: Tensor, keep_mask: Optional[Tensor], p: float) -> Tensor:
    """nn.Dropout (layers.py:248,256) with the Bernoulli draw made explicit.

    ``keep_mask`` None  -> eval mode (identity).  Otherwise a {0,1} tensor of x's
    logical shape; kept elements are scaled by 1/(1-p) (inverted dropout).
    """
    if keep_mask is None:
        return x
    if p >= 1.0:
        return x * 0
    return x * keep_mask.to(x.dtype) * (1.0 / (1.0 - p))


def st_conv_block(x: Tensor, gso: Tensor, p: Dict[str, Tensor], prefix: str, cfg: OracleConfig,
                  last_block_channel: int, channels: Sequence[int],
                  keep_mask: Optional[Tensor] = None) -> Tensor:
    """model/layers.py:250-258 -- T G (relu) T N D."""
    n_vertex = x.shape[-1]
    x = temporal_conv(x, p, prefix + "tmp_conv1.", cfg.Kt, last_block_channel, channels[0], cfg.act_func)  # :251
    x = graph_conv_layer(x, gso, p, prefix + "graph_conv.", cfg.graph_conv_type, channels[0], channels[1])  # :252
    x = torch.relu(x)                                               # layers.py:253
    x = temporal_conv(x, p, prefix + "tmp_conv2.", cfg.Kt, channels[1], channels[2], cfg.act_func)         # :254
    x = F.layer_norm(x.permute(0, 2, 3, 1), [n_vertex, channels[2]], p[prefix + "tc2_ln.weight"],
                     p[prefix + "tc2_ln.bias"], eps=1e-12).permute(0, 3, 1, 2)   # layers.py:246,255
    return dropout(x, keep_mask, cfg.droprate)                      # layers.py:256


def output_block(x: Tensor, p: Dict[str, Tensor], prefix: str, cfg: OracleConfig, Ko: int,
                 last_block_channel: int, channels: Sequence[int], end_channel: int,
                 keep_mask: Optional[Tensor] = None) -> Tensor:
    """model/layers.py:276-284 -- T N F (relu, dropout) F."""
    n_vertex = x.shape[-1]
    x = temporal_conv(x, p, prefix + "tmp_conv1.", Ko, last_block_channel, channels[0], cfg.act_func)  # :277
    x = F.layer_norm(x.permute(0, 2, 3, 1), [n_vertex, channels[0]], p[prefix + "tc1_ln.weight"],
                     p[prefix + "tc1_ln.bias"], eps=1e-12)          # layers.py:272,278
    x = F.linear(x, p[prefix + "fc1.weight"], p.get(prefix + "fc1.bias"))   # layers.py:279
    x = torch.relu(x)                                               # layers.py:280
    x = dropout(x, keep_mask, cfg.droprate)                         # layers.py:281 (mask shape (B,1,N,channels[1]))
    x = F.linear(x, p[prefix + "fc2.weight"], p.get(prefix + "fc2.bias")).permute(0, 3, 1, 2)  # :282
    return x


def stgcn_forward(x: Tensor, gso: Tensor, p: Dict[str, Tensor], cfg: OracleConfig,
                  keep_masks: Optional[Sequence[Optional[Tensor]]] = None,
                  return_block_outputs: bool = False):
    """model/models.py:44-53 (STGCNChebGraphConv) / :94-103 (STGCNGraphConv).

    ``keep_masks``: one entry per dropout site in forward order
    (st_blocks.0, st_blocks.1, ..., output) or None for eval mode.
    """
    blocks = cfg.blocks
    outs = []
    n_st = cfg.n_st_blocks
    for l in range(n_st):                                           # models.py:31-33
        km = None if keep_masks is None else keep_masks[l]
        x = st_conv_block(x, gso, p, f"st_blocks.{l}.", cfg, blocks[l][-1], blocks[l + 1], km)
        outs.append(x)
    Ko = cfg.Ko
    if Ko > 1:                                                      # models.py:46-47
        km = None if keep_masks is None else keep_masks[n_st]
        x = output_block(x, p, "output.", cfg, Ko, blocks[-3][-1], blocks[-2], blocks[-1][0], km)
    elif Ko == 0:                                                   # models.py:48-51 (no dropout applied in fwd)
        x = F.linear(x.permute(0, 2, 3, 1), p["fc1.weight"], p.get("fc1.bias"))
        x = torch.relu(x)
        x = F.linear(x, p["fc2.weight"], p.get("fc2.bias")).permute(0, 3, 1, 2)
    # Ko == 1: the reference silently skips the head (models.py:46-51)
    if return_block_outputs:
        return x, outs
    return x


# --------------------------------------------------------------------------- parameters
def param_shapes(cfg: OracleConfig, n_vertex: int) -> Dict[str, tuple]:
    """state_dict keys and shapes of the reference model.

    Follows the constructors: Align (layers.py:12) always allocates align_conv,
    CausalConv2d (layers.py:98) has 2*c_out outputs for glu/gtu, ChebGraphConv
    weight (Ks,c,c) (layers.py:129), GraphConv weight (c,c) (layers.py:180),
    LayerNorm([N,C]) (layers.py:246,272), Linear (layers.py:270-271).
    """
    shapes: Dict[str, tuple] = {}
    gated = cfg.act_func in ("glu", "gtu")

    def tconv(prefix, Kt, c_in, c_out):
        shapes[prefix + "align.align_conv.weight"] = (c_out, c_in, 1, 1)
        shapes[prefix + "align.align_conv.bias"] = (c_out,)
        co = 2 * c_out if gated else c_out
        shapes[prefix + "causal_conv.weight"] = (co, c_in, Kt, 1)
        shapes[prefix + "causal_conv.bias"] = (co,)

    blocks = cfg.blocks
    for l in range(cfg.n_st_blocks):
        pre = f"st_blocks.{l}."
        c_last, ch = blocks[l][-1], blocks[l + 1]
        tconv(pre + "tmp_conv1.", cfg.Kt, c_last, ch[0])
        shapes[pre + "graph_conv.align.align_conv.weight"] = (ch[1], ch[0], 1, 1)
        shapes[pre + "graph_conv.align.align_conv.bias"] = (ch[1],)
        if cfg.graph_conv_type == "cheb_graph_conv":
            shapes[pre + "graph_conv.cheb_graph_conv.weight"] = (cfg.Ks, ch[1], ch[1])
            if cfg.enable_bias:
                shapes[pre + "graph_conv.cheb_graph_conv.bias"] = (ch[1],)
        else:
            shapes[pre + "graph_conv.graph_conv.weight"] = (ch[1], ch[1])
            if cfg.enable_bias:
                shapes[pre + "graph_conv.graph_conv.bias"] = (ch[1],)
        tconv(pre + "tmp_conv2.", cfg.Kt, ch[1], ch[2])
        shapes[pre + "tc2_ln.weight"] = (n_vertex, ch[2])
        shapes[pre + "tc2_ln.bias"] = (n_vertex, ch[2])
    Ko = cfg.Ko
    if Ko > 1:
        tconv("output.tmp_conv1.", Ko, blocks[-3][-1], blocks[-2][0])
        shapes["output.fc1.weight"] = (blocks[-2][1], blocks[-2][0])
        if cfg.enable_bias:
            shapes["output.fc1.bias"] = (blocks[-2][1],)
        shapes["output.fc2.weight"] = (blocks[-1][0], blocks[-2][1])
        if cfg.enable_bias:
            shapes["output.fc2.bias"] = (blocks[-1][0],)
        shapes["output.tc1_ln.weight"] = (n_vertex, blocks[-2][0])
        shapes["output.tc1_ln.bias"] = (n_vertex, blocks[-2][0])
    elif Ko == 0:
        shapes["fc1.weight"] = (blocks[-2][0], blocks[-3][-1])
        if cfg.enable_bias:
            shapes["fc1.bias"] = (blocks[-2][0],)
        shapes["fc2.weight"] = (blocks[-1][0], blocks[-2][0])
        if cfg.enable_bias:
            shapes["fc2.bias"] = (blocks[-1][0],)
    return shapes


def random_params(cfg: OracleConfig, n_vertex: int, seed: int = 0, dtype=torch.float32,
                  scale: float = 1.0) -> Dict[str, Tensor]:
    """Deterministic synthetic parameters (NOT the reference's init distribution --
    for parity of the *arithmetic* any well-scaled values do).  fan-in scaled
    uniform for weights, small uniform for biases, LN gamma around 1.

    Drawn from ``numpy.random.RandomState`` (legacy MT19937 stream, frozen by
    numpy's compatibility policy) in float64, in ``param_shapes`` order, so that
    fixtures need not store the parameters themselves (checksums are stored)."""
    import numpy as np
    rs = np.random.RandomState(seed)
    out: Dict[str, Tensor] = {}
    for name, shp in param_shapes(cfg, n_vertex).items():
        u = rs.uniform(-1.0, 1.0, size=shp)
        if "_ln.weight" in name:
            t = 1.0 + 0.1 * u
        elif "_ln.bias" in name or name.endswith("bias"):
            t = 0.1 * u
        else:
            fan_in = 1
            for d in shp[1:]:
                fan_in *= d
            if name.endswith("cheb_graph_conv.weight"):
                fan_in = shp[1]
            if name.endswith("graph_conv.graph_conv.weight"):
                fan_in = shp[0]
            t = scale * math.sqrt(3.0 / max(fan_in, 1)) * u
        out[name] = torch.from_numpy(np.ascontiguousarray(t)).to(dtype)
    return out


def param_checksums(p: Dict[str, Tensor]):
    """(sum, sum|.|) over all parameters in float64 -- stored in fixtures to detect RNG drift."""
    s = sum(float(v.double().sum()) for v in p.values())
    a = sum(float(v.double().abs().sum()) for v in p.values())
    return s, a


# --------------------------------------------------------------------------- training step
def mse_loss(y_pred: Tensor, y: Tensor) -> Tensor:
    """nn.MSELoss() default 'mean' reduction (main.py:136, used at :166-167)."""
    return ((y_pred.reshape(y_pred.shape[0], -1) - y) ** 2).mean()


def loss_and_grads(x: Tensor, y: Tensor, gso: Tensor, p: Dict[str, Tensor], cfg: OracleConfig,
                   keep_masks=None):
    """fwd + MSE + bwd of main.py:165-168 through autograd on the restated forward.

    Returns (loss, {name: grad or None}).  Parameters that never receive a
    gradient in the reference (unused ``align_conv`` tensors)
    come back as None, mirroring ``param.grad is None``.
    """
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
    out = stgcn_forward(x, gso, leaves, cfg, keep_masks)
    loss = mse_loss(out, y)
    names = list(leaves.keys())
    grads = torch.autograd.grad(loss, [leaves[n] for n in names], allow_unused=True)
    return loss.detach(), {n: g for n, g in zip(names, grads)}


def adamw_step(p: Dict[str, Tensor], grads: Dict[str, Optional[Tensor]], state: Dict[str, dict],
               lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-3) -> None:
    """torch.optim.AdamW single-tensor update as configured at main.py:148
    (amsgrad False, maximize False).  Parameters whose grad is None are skipped
    entirely -- no decay, no step count (torch/optim/adamw.py behaviour)."""
    b1, b2 = betas
    for name, w in p.items():
        g = grads.get(name)
        if g is None:
            continue
        st = state.setdefault(name, {"step": 0, "m": torch.zeros_like(w), "v": torch.zeros_like(w)})
        st["step"] += 1
        t = st["step"]
        w.mul_(1.0 - lr * weight_decay)
        st["m"].mul_(b1).add_(g, alpha=1.0 - b1)
        st["v"].mul_(b2).addcmul_(g, g, value=1.0 - b2)
        bc1 = 1.0 - b1 ** t
        bc2 = 1.0 - b2 ** t
        denom = (st["v"].sqrt() / math.sqrt(bc2)).add_(eps)
        w.addcdiv_(st["m"], denom, value=-lr / bc1)


def train_step(x: Tensor, y: Tensor, gso: Tensor, p: Dict[str, Tensor], cfg: OracleConfig,
               opt_state: Dict[str, dict], keep_masks=None, lr: float = 1e-3, weight_decay: float = 1e-3):
    """One iteration of the loop body main.py:165-169 (zero_grad, fwd, loss, bwd, AdamW)."""
    loss, grads = loss_and_grads(x, y, gso, p, cfg, keep_masks)
    adamw_step(p, grads, opt_state, lr=lr, weight_decay=weight_decay)
    return loss, grads


# --------------------------------------------------------------------------- metrics
def evaluate_metric_arrays(y_true, y_pred):
    """script/utility.py:103-121 on already inverse-transformed flat arrays:
    MAE, RMSE, WMAPE = sum|d| / sum(y)."""
    import numpy as np
    d = np.abs(y_true - y_pred)
    return float(d.mean()), float(np.sqrt((d ** 2).mean())), float(d.sum() / y_true.sum())
