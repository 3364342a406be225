"""bf16 configurations (BASELINE.json configs[2], [4]): run one ST block with bfloat16 activations through the bound library (the CPU
emulator or the HIP library, same host code) and compare every stored tensor and every gradient with the bf16 statement of the stage
oracle (oracle/stblock_stages.py, QuantBf16: same rounding points, float64 accumulation).

Tolerances (stated here, asserted by `assert_bf16_errors`):
  * stored bf16 tensors: the HIP value and the oracle value are both bf16 numbers formed from sums that differ only in accumulation
    order / precision (fp32 vs float64), so they are EQUAL except where the sum sits on a rounding boundary (one bf16 ulp apart), and a
    flipped value then perturbs what is computed from it.  Bars: relative rms error <= 2^-9 (0.2 %: well below the bf16 quantisation
    noise of 2^-9 / sqrt(3) per element that the fp32 -> bf16 statement itself introduces) and max |diff| <= 2^-5 of the tensor's max.
  * fp32 outputs (LayerNorm statistics, parameter gradients): relative-to-max error <= 1e-2 (parameter gradients are sums over
    10^4 - 10^5 products of bf16 numbers: flips average out; measured values are ~1e-3).
"""
import ctypes as C

import numpy as np
import torch

from oracle import stblock_stages as st
from stgcn_amd import _lib, ops
from tests.emu_util import block_case, nonsym_gso, params_in_field_order

Q = st.QuantBf16()
RMS_TOL = 2.0 ** -9
MAX_TOL = 2.0 ** -5
F32_TOL = 1e-2


def bf16_tensor(a: np.ndarray, dev) -> torch.Tensor:
    """float array -> torch.bfloat16 tensor holding RNE-rounded values."""
    bits = st.to_bf16_bits(a).astype(np.int16)
    return torch.from_numpy(bits).view(torch.bfloat16).to(dev)


def bf16_numpy(t: torch.Tensor) -> np.ndarray:
    return st.from_bf16_bits(t.detach().cpu().contiguous().view(torch.int16).numpy().view(np.uint16))


def seg_bf16(buf_f32: np.ndarray, off_floats: int, shape) -> np.ndarray:
    """A bf16 tensor stored inside a float32 plan buffer (offsets count 4-byte units)."""
    n = int(np.prod(shape))
    bits = buf_f32.view(np.uint16)[2 * off_floats:2 * off_floats + n]
    return st.from_bf16_bits(bits).reshape(shape)


def err_stored(got, ref):
    ref = np.asarray(ref, dtype=np.float64)
    d = np.abs(np.asarray(got, dtype=np.float64) - ref)
    rms_ref = max(1e-30, float(np.sqrt((ref ** 2).mean())))
    return dict(rms=float(np.sqrt((d ** 2).mean())) / rms_ref, max=float(d.max()) / max(1e-30, float(np.abs(ref).max())),
                mismatch=float((d > 0).mean()))


def run_block_case_bf16(dev, c_in, channels, Kt, Ks, gct, act, N, B, T, training, gso=None, seed=99, offset=3, pdrop=0.5, gc_form=None,
                        ln_scale=None, consumer=None):
    """Returns ({name: stored-tensor error dict}, {name: relative-to-max error of an fp32 output}).
    gc_form: "poly" (slab-resident graph conv) / "recursion" (tiled path); None = whichever the plan selects.
    ln_scale = (g, b): the block's LayerNorm parameters are drawn as gamma = g * U(-1, 1), beta = b * U(-1, 1) (the
    backward rebuilds sum g * xhat from dy * (y - keep_scale * beta) with y in bf16, which amplifies y's rounding by |beta| / |gamma xhat|).
    consumer = "block" / "head": the block's output feeds a second fused module and dy is what THAT module's backward produces, so the
    LayerNorm-backward row partials come out of the consumer's hook epilogue (tc1_bwd_kernel / the head's transposed conv) instead of
    ln_bwd_rowstats_kernel; everything is then checked as for an external dy."""
    L = _lib.lib()
    cfg, p = block_case(c_in, channels, Kt, Ks, gct, act, N, B, T)
    if ln_scale is not None:
        rl = np.random.RandomState(77)
        shp = tuple(p["st_blocks.0.tc2_ln.weight"].shape)
        p["st_blocks.0.tc2_ln.weight"] = torch.from_numpy((ln_scale[0] * rl.uniform(-1, 1, shp)).astype(np.float32))
        p["st_blocks.0.tc2_ln.bias"] = torch.from_numpy((ln_scale[1] * rl.uniform(-1, 1, shp)).astype(np.float32))
    if gso is None:
        gso = nonsym_gso(N, 5)
    rs = np.random.RandomState(11)
    x_np = Q(rs.standard_normal((B, c_in, T, N)))                     # bf16 values, logical NCHW
    T2 = T - 2 * (Kt - 1)
    dy_np = Q(rs.standard_normal((B, channels[2], T2, N)))
    bcfg = ops.BlockConfig(Kt=Kt, Ks=Ks, n_vertex=N, c_in=c_in, channels=tuple(channels), act_func=act, graph_conv_type=gct, droprate=pdrop)
    gp, gt = ops.gso_prepare(torch.from_numpy(gso).to(dev), ops.graph_terms(bcfg))
    params = [None if t is None else t.clone().to(dev).requires_grad_(True) for t in params_in_field_order(p, "st_blocks.0.", gct)]
    x = bf16_tensor(x_np, dev).requires_grad_(c_in > 1)
    wsc = ops.WorkspaceCache()
    y = ops.st_conv_block(x, gp, gt, bcfg, params, training, seed, offset, wsc)
    assert y.dtype == torch.bfloat16
    hooked = None
    if consumer is None:
        y.backward(bf16_tensor(dy_np, dev))
    else:
        y.retain_grad()
        L.dll.stgcn_profile_enable(1)
        if consumer == "block":
            _, pc = block_case(channels[2], channels, Kt, Ks, gct, act, N, B, T2, seed=8)
            ccfg = ops.BlockConfig(Kt=Kt, Ks=Ks, n_vertex=N, c_in=channels[2], channels=tuple(channels), act_func=act, graph_conv_type=gct, droprate=pdrop)
            cparams = [None if t is None else t.clone().to(dev).requires_grad_(True) for t in params_in_field_order(pc, "st_blocks.0.", gct)]
            out = ops.st_conv_block(y, gp, gt, ccfg, cparams, training, seed + 1, offset, ops.WorkspaceCache())
            out.backward(bf16_tensor(Q(rs.standard_normal(tuple(out.shape))), dev))
        else:
            from oracle import stgcn_oracle as orc
            hc = orc.OracleConfig(Kt=Kt, Ks=Ks, n_his=12, act_func=act, graph_conv_type=gct, droprate=pdrop,
                                  blocks=[[1], [64, 16, 64], [64, 16, channels[2]], [128, 128], [1]])
            hp_ = {k: v for k, v in orc.random_params(hc, N, seed=3, dtype=torch.float32).items() if k.startswith("output.")}
            hn = ["tmp_conv1.causal_conv.weight", "tmp_conv1.causal_conv.bias", "tmp_conv1.align.align_conv.weight", "tmp_conv1.align.align_conv.bias",
                  "tc1_ln.weight", "tc1_ln.bias", "fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias"]
            assert tuple(hp_["output.tmp_conv1.causal_conv.weight"].shape)[2] >= 1
            hparams = [hp_["output." + n].clone().to(dev).requires_grad_(True) for n in hn]
            Ko = tuple(hp_["output.tmp_conv1.causal_conv.weight"].shape)[2]
            assert Ko == T2, (Ko, T2)       # (the head consumes the whole remaining time axis: T - 2 (Kt - 1) must be the fixture's Ko)
            hcfg = ops.HeadConfig(Ko=Ko, n_vertex=N, c_in=channels[2], channels=(128, 128), end_channel=1, act_func=act, droprate=pdrop)
            out = ops.output_block(y, hcfg, hparams, training, seed + 1, offset, ops.WorkspaceCache())
            out.backward(torch.from_numpy(rs.standard_normal(tuple(out.shape)).astype(np.float32)).to(dev))
        buf = C.create_string_buffer(1 << 16)
        L.dll.stgcn_profile_collect(buf, len(buf))
        L.dll.stgcn_profile_enable(0)
        import json
        launched = json.loads(buf.value.decode())
        n_rs = sum(v["calls"] for k, v in launched.items() if k.startswith("ln_bwd_rowstats"))
        hooked = n_rs == (1 if consumer == "block" else 0)      # (a consumer BLOCK runs the pass for its own LayerNorm: its dy came from outside)
        dy_np = bf16_numpy(y.grad)
    if str(dev).startswith("cuda"):
        torch.cuda.synchronize()

    cl = lambda a: np.ascontiguousarray(a.transpose(0, 2, 3, 1)).astype(np.float64)
    keep = None
    if training:
        ks = ops.dropout_mask(B * T2 * N * channels[2], pdrop, seed, offset, dev).cpu().numpy().reshape(B, T2, N, channels[2])
        keep = (ks > 0).astype(np.float64)
    g64 = gso.astype(np.float64)
    bp = st.block_params_np(p, "st_blocks.0.", gct, np.float64)
    desc = ops.make_desc(bcfg, B, T, training, c_in > 1, dtype=torch.bfloat16)
    plan = ops.query_plan(desc)
    if gc_form is None:
        gc_form = "recursion" if plan.tiled_gc else "poly"
    y_ref, sv = st.stblock_fwd(cl(x_np), g64, bp, Kt, c_in, channels, gct, act, keep, pdrop, q=Q, gc_form=gc_form, ln_from_stored=N > 448)
    # the same block in the fp64 statement: how far the bf16 configuration is from the reference arithmetic (reported, loosely bounded)
    y64, sv64 = st.stblock_fwd(cl(x_np), g64, bp, Kt, c_in, channels, gct, act, keep, pdrop)

    ws = wsc.buf.cpu().numpy()
    saved = torch.empty(plan.saved_floats, device=dev)
    y2 = torch.empty(B, T2, N, channels[2], dtype=torch.bfloat16, device=dev)
    pst = ops._param_struct(_lib.StblockParams, [None if t is None else t.detach() for t in params])
    x_cl = x.detach().permute(0, 2, 3, 1).contiguous()
    ws2 = torch.empty(plan.ws_floats, device=dev)
    stream = torch.cuda.current_stream().cuda_stream if str(dev).startswith("cuda") else None
    L.check(L.dll.stgcn_stblock_forward(C.byref(desc), C.byref(pst), x_cl.data_ptr(), gp.data_ptr(), y2.data_ptr(), saved.data_ptr(),
                                        ws2.data_ptr(), seed, offset, None, stream), "fwd")
    if str(dev).startswith("cuda"):
        torch.cuda.synchronize()
    svn = saved.cpu().numpy()
    T1 = plan.T1
    c0, c1, c2 = channels
    terms = 2 if gct == "graph_conv" else Ks
    stored, f32 = {}, {}
    if hooked is not None:
        f32["grad_none_ok.hook_epilogue_used"] = 0.0 if hooked else 1.0      # (tolerance 0: the consumer's epilogue wrote the row partials, no ln_bwd_rowstats launch)
    if not plan.recompute_tc1:
        stored["fwd.U1"] = err_stored(seg_bf16(svn, plan.sv_U1, (B, T1, N, c0)), sv["U1"])
        stored["fwd.S1"] = err_stored(seg_bf16(svn, plan.sv_S1, (B, T1, N, c0)), sv["S1"])
    stored["fwd.A"] = err_stored(seg_bf16(svn, plan.sv_A, (B, T1, N, c1)), sv["A"])
    per = B * T1 * N * c1 // 2      # 4-byte units per term: the X_k follow each other without padding (c1 = 16 elements per row)
    for k in range(1, terms):
        stored[f"fwd.X{k}"] = err_stored(seg_bf16(svn, plan.sv_Xk + (k - 1) * per, (B, T1, N, c1)), sv["Xs"][k])
    stored["fwd.G"] = err_stored(seg_bf16(svn, plan.sv_G, (B, T1, N, c1)), sv["G"])
    if plan.stored_US2:      # (only when LayerNorm runs as a separate pass over the stored gate inputs: more than 448 nodes)
        stored["fwd.U2"] = err_stored(seg_bf16(svn, plan.sv_U2, (B, T2, N, c2)), Q(sv["U2"]))
        stored["fwd.S2"] = err_stored(seg_bf16(svn, plan.sv_S2, (B, T2, N, c2)), Q(sv["S2"]))
    stored["fwd.y"] = err_stored(cl(bf16_numpy(y)), y_ref)
    stored["fwd.y_vs_fp64"] = err_stored(cl(bf16_numpy(y)), y64)          # (reported; bounded loosely below)
    f32["fwd.mean"] = float(np.abs(svn[plan.sv_mean:plan.sv_mean + B * T2].reshape(B, T2) - sv["mean"]).max() / max(1e-30, np.abs(sv["mean"]).max() + 1e-3))
    f32["fwd.rstd"] = float(np.abs(svn[plan.sv_rstd:plan.sv_rstd + B * T2].reshape(B, T2) / sv["rstd"] - 1).max())
    f32["fwd.y_repeat_bitwise"] = float((y2.view(torch.int16) != y.detach().permute(0, 2, 3, 1).contiguous().view(torch.int16)).sum().item())
    # Backward: checked on the HIP path's OWN saved tensors ("teacher forcing").  ReLU is discontinuous: where a stored G sits on the
    # rounding boundary next to zero the two forwards may disagree about the mask, and a flipped mask changes the gradient entering the
    # graph conv by O(1) in that element -- a property of the forward's rounding, not of the backward kernels under test.
    svh = dict(sv)
    if not plan.recompute_tc1:
        svh["U1"], svh["S1"] = seg_bf16(svn, plan.sv_U1, (B, T1, N, c0)), seg_bf16(svn, plan.sv_S1, (B, T1, N, c0))
        svh["H1"] = st._gate(svh["U1"], svh["S1"], act)
    svh["A"] = seg_bf16(svn, plan.sv_A, (B, T1, N, c1))
    svh["Xs"] = [svh["A"]] + [seg_bf16(svn, plan.sv_Xk + (k - 1) * per, (B, T1, N, c1)) for k in range(1, terms)]
    svh["G"] = seg_bf16(svn, plan.sv_G, (B, T1, N, c1))
    svh["U2"], svh["S2"], svh["H2"] = st.tconv_fwd(svh["G"], sv["W2"], st.fold_tconv(bp["tc2_w"], bp["tc2_b"], bp["tc2_aw"], bp["tc2_ab"], c1, c2, Kt)[1],
                                                   Kt, c2, act, Q)      # what tc2_bwd_kernel recomputes from the stored G
    svh["y"] = cl(bf16_numpy(y))
    svh["mean"] = svn[plan.sv_mean:plan.sv_mean + B * T2].reshape(B, T2).astype(np.float64)
    svh["rstd"] = svn[plan.sv_rstd:plan.sv_rstd + B * T2].reshape(B, T2).astype(np.float64)
    stages = {}
    dx_ref, g_ref = st.stblock_bwd(cl(dy_np), svh, g64, bp, Kt, c_in, channels, gct, act, pdrop, need_dx=c_in > 1, q=Q, gc_form=gc_form,
                                   stages=stages)
    # How much of the backward the teacher forcing hides: the ReLU-mask flip rate between the two forwards, and the
    # gradients against the PURE oracle (its own saved tensors, no forcing), bounded in rms: a flipped mask element changes the gradient
    # entering the graph conv by O(1) there, so the un-forced error is ~sqrt(flip rate) in rms, not a kernel defect.
    flip = float(((svh["G"] > 0) != (np.asarray(sv["G"]) > 0)).mean())
    f32["relu_flip_rate"] = flip
    dx_u, g_u = st.stblock_bwd(cl(dy_np), sv, g64, bp, Kt, c_in, channels, gct, act, pdrop, need_dx=c_in > 1, q=Q, gc_form=gc_form)
    rms = lambda a: float(np.sqrt((np.asarray(a, np.float64) ** 2).mean()))
    if c_in > 1:
        f32["unforced_rms.dx"] = rms(cl(bf16_numpy(x.grad)) - dx_u) / max(1e-30, rms(dx_u))
    worst = 0.0
    for name, prm in zip(_lib.PARAM_FIELDS, params):
        if prm is not None and prm.grad is not None and g_u.get(name) is not None:
            worst = max(worst, rms(prm.grad.cpu().numpy().astype(np.float64) - g_u[name].reshape(prm.shape)) / max(1e-30, rms(g_u[name])))
    f32["unforced_rms.param_grads_worst"] = worst
    stored["bwd.dYg"] = err_stored(seg_bf16(ws, plan.ws_dYg, (B, T1, N, c1)), stages["dYg"])
    stored["bwd.dA"] = err_stored(seg_bf16(ws, plan.ws_dA, (B, T1, N, c1)), stages["dA"])
    if c_in > 1:
        stored["bwd.dx"] = err_stored(cl(bf16_numpy(x.grad)), dx_ref)
    rel = lambda got, ref: float(np.abs(got - ref).max() / max(1e-30, np.abs(ref).max()))
    for name, prm in zip(_lib.PARAM_FIELDS, params):
        ref = g_ref[name]
        if prm is None:
            continue
        if ref is None:
            f32["grad_none_ok." + name] = 0.0 if prm.grad is None else 1.0
            continue
        f32["grad." + name] = rel(prm.grad.cpu().numpy().astype(np.float64), ref.reshape(prm.shape)) if prm.grad is not None else float("inf")
    return stored, f32


def assert_bf16_errors(stored, f32):
    import json, os
    if os.environ.get("STGCN_BF16_REPORT"):      # (GPU passes: the measured flip rates / un-forced errors go into profiles/)
        with open(os.environ["STGCN_BF16_REPORT"], "a") as fh:
            fh.write(json.dumps({"f32": {k: v for k, v in f32.items() if "flip" in k or "unforced" in k},
                                 "stored_rms": {k: round(v["rms"], 6) for k, v in stored.items()}}) + "\n")
    bad = {}
    for k, e in stored.items():
        if k.endswith("_vs_fp64"):
            if not (e["rms"] <= 3e-2):          # bf16 storage + bf16 operands vs exact arithmetic: ~1 % rms on unit-variance outputs
                bad[k] = e
            continue
        if not (e["rms"] <= RMS_TOL and e["max"] <= MAX_TOL):
            bad[k] = e
    for k, v in f32.items():
        # (LayerNorm statistics are fp32 sums, but of values downstream of bf16 tensors in which a few elements differ by one ulp)
        # relu_flip_rate: measured 0 - 3e-4 (elements of G within one bf16 ulp of zero); un-forced gradients: rms error <= 5 % (measured ~1 %)
        tol = (0.0 if (k.startswith("grad_none_ok") or k.endswith("bitwise")) else 1e-3 if k in ("fwd.mean", "fwd.rstd") else
               2e-3 if k == "relu_flip_rate" else 5e-2 if k.startswith("unforced_rms") else F32_TOL)
        if not (v <= tol):
            bad[k] = v
    assert not bad, f"out of tolerance: {bad}\nstored: {stored}\nf32: {f32}"


def run_head_case_bf16(dev, N, B, c_in=64, channels=(128, 128), Ko=4, act="glu", training=True, seed=5, offset=9, pdrop=0.5):
    """The output head with bf16 activations against outblock_fwd / outblock_bwd of the stage oracle (q = QuantBf16)."""
    from oracle import stgcn_oracle as orc
    cfg = orc.OracleConfig(Kt=3, Ks=3, n_his=12, act_func=act, graph_conv_type="cheb_graph_conv", droprate=pdrop,
                           blocks=[[1], [64, 16, 64], [64, 16, c_in], list(channels), [1]])
    full = orc.random_params(cfg, N, seed=3, dtype=torch.float32)
    p = {k: v for k, v in full.items() if k.startswith("output.")}
    names = ["tmp_conv1.causal_conv.weight", "tmp_conv1.causal_conv.bias", "tmp_conv1.align.align_conv.weight", "tmp_conv1.align.align_conv.bias",
             "tc1_ln.weight", "tc1_ln.bias", "fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias"]
    params = [p["output." + n].clone().to(dev).requires_grad_(True) for n in names]
    rs = np.random.RandomState(4)
    x_np = Q(rs.standard_normal((B, c_in, Ko, N)))
    dout_np = rs.standard_normal((B, 1, 1, N)).astype(np.float32)
    hcfg = ops.HeadConfig(Ko=Ko, n_vertex=N, c_in=c_in, channels=tuple(channels), end_channel=1, act_func=act, droprate=pdrop)
    x = bf16_tensor(x_np, dev).requires_grad_(True)
    wsc = ops.WorkspaceCache()
    out = ops.output_block(x, hcfg, params, training, seed, offset, wsc)
    assert out.dtype == torch.float32 and out.shape == (B, 1, 1, N)
    out.backward(torch.from_numpy(dout_np).to(dev))
    if str(dev).startswith("cuda"):
        torch.cuda.synchronize()
    cl = lambda a: np.ascontiguousarray(a.transpose(0, 2, 3, 1)).astype(np.float64)
    keep = None
    c0, c1 = channels
    if training:
        ks = ops.dropout_mask(B * N * c1, pdrop, seed, offset, dev).cpu().numpy().reshape(B, 1, N, c1)
        keep = (ks > 0).astype(np.float64)
    hp = st.head_params_np(p, np.float64)
    out_ref, sv = st.outblock_fwd(cl(x_np), hp, Ko, c_in, channels, act, keep, pdrop, q=Q)
    dx_ref, g_ref = st.outblock_bwd(dout_np[:, 0].astype(np.float64), sv, hp, Ko, c_in, channels, act, pdrop, True, q=Q)
    rel = lambda got, ref: float(np.abs(got - ref).max() / max(1e-30, np.abs(ref).max()))
    stored = {"head.dx": err_stored(cl(bf16_numpy(x.grad)), dx_ref)}
    f32 = {"head.out": rel(out.detach().cpu().numpy()[:, 0].astype(np.float64), out_ref)}
    keys = ["tc_w", "tc_b", "tc_aw", "tc_ab", "ln_w", "ln_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b"]
    for k, prm in zip(keys, params):
        ref = g_ref[k]
        if ref is None:
            f32["grad_none_ok.head." + k] = 0.0 if prm.grad is None else 1.0
        else:
            f32["grad.head." + k] = rel(prm.grad.cpu().numpy().astype(np.float64), np.asarray(ref).reshape(prm.shape)) if prm.grad is not None else float("inf")
    return stored, f32
