"""Stage-level CPU restatement (numpy, explicit backward) of one ST-Conv block.

TEST INFRASTRUCTURE ONLY (see oracle/stgcn_oracle.py header for the import rule).

Where stgcn_oracle.py restates the reference *module by module* and lets autograd
differentiate, this file restates the same arithmetic *kernel by kernel*, in the
decomposition the HIP path uses, with hand-written backward
passes.  It is validated against stgcn_oracle.py (hence against the reference)
in tests/test_stage_oracle.py, and is then the checker for each HIP kernel.

Layout: every activation is channels-last ``(B, T, N, C)`` (the memory layout the
reference itself ends up in after its first graph conv).

Formulas follow SURVEY.md section 8a rows a2/a4/a6 (which cite model/layers.py):
  temporal conv + GLU   layers.py:87-105     graph conv  layers.py:143-172, 194-206, 222-231
  LayerNorm + dropout   layers.py:246-256
"""
from __future__ import annotations

import numpy as np


# ----------------------------------------------------------------------------- bf16 statement
class QuantBf16:
    """Rounding rule of the bf16 configurations (BASELINE.json configs[2], [4]; include/stgcn_hip.h STGCN_DTYPE_BF16), as a callable
    ``q(a)``: the value is first an fp32 number (what the kernels hold in registers / LDS), then rounded to bfloat16 with
    round-to-nearest-even and returned as float64.  The HIP path applies it at exactly two kinds of places, and every function of
    this file that takes ``q`` applies it at the same places:

      (1) every activation / saved tensor / activation gradient that crosses a kernel boundary in HBM
          (x, U, S, A, X_k, G, y; dy, dYg, dA, dx; the head's yln, hd, dh1, dyln, dZ) -- "stored" below;
      (2) both operands of every matrix product (im2col tiles, weights, operator polynomials, dZ tiles, G_k, H) where they enter
          the matrix cores; accumulation is fp32 (float64 here).

    Everything else is NOT rounded: accumulators, biases, gate math, LayerNorm statistics / gamma / beta, dropout scaling, parameter
    gradients and their partial sums, the optimizer.  ``q = None`` (the default everywhere) is the fp32 / fp64 statement."""

    def __call__(self, a):
        a32 = np.ascontiguousarray(np.asarray(a, dtype=np.float32))
        u = a32.view(np.uint32).astype(np.uint64)
        u = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
        return u.astype(np.uint32).view(np.float32).astype(np.float64).reshape(np.shape(a))


def _q(q, a):
    return a if q is None else q(a)


def to_bf16_bits(a):
    """float array -> uint16 bfloat16 bit patterns (RNE), e.g. to hand test inputs to the HIP path."""
    a32 = np.ascontiguousarray(np.asarray(a, dtype=np.float32))
    u = a32.view(np.uint32).astype(np.uint64)
    return (((u + 0x7FFF + ((u >> 16) & 1)) >> 16) & 0xFFFF).astype(np.uint16).reshape(np.shape(a))


def from_bf16_bits(b):
    """uint16 bfloat16 bit patterns -> float64."""
    b = np.ascontiguousarray(np.asarray(b, dtype=np.uint16))
    return (b.astype(np.uint32) << 16).view(np.float32).astype(np.float64).reshape(b.shape)


# ----------------------------------------------------------------------------- folding
def fold_tconv(conv_w, conv_b, align_w, align_b, c_in, c_out, Kt, gated=True):
    """Fold the residual branch ``Align(x)[:, :, Kt-1:]`` (layers.py:88, 14-23) into the
    causal-conv weights (layers.py:89): the residual only touches tap Kt-1 and the P half.

    Returns W_eff [Kt*c_in, co] (row index = tap*c_in + i) and b_eff [co], co = 2*c_out if gated.
      c_in >  c_out : R = align_conv(x)            -> W_eff[(Kt-1)*c_in+i, o] += align_w[o,i], b_eff[o] += align_b[o]
      c_in <= c_out : R = x zero-padded / identity -> W_eff[(Kt-1)*c_in+i, i] += 1
    """
    co = conv_w.shape[0]
    W = np.zeros((Kt * c_in, co), dtype=conv_w.dtype)
    for k in range(Kt):
        W[k * c_in:(k + 1) * c_in, :] = conv_w[:, :, k, 0].T
    b = np.zeros(co, dtype=conv_w.dtype) if conv_b is None else conv_b.copy()
    base = (Kt - 1) * c_in
    if c_in > c_out:
        W[base:base + c_in, :c_out] += align_w[:, :, 0, 0].T
        if align_b is not None:
            b[:c_out] += align_b
    else:
        for i in range(min(c_in, c_out)):
            W[base + i, i] += 1.0
    return W, b


def unfold_tconv_grads(dW_eff, db_eff, c_in, c_out, Kt):
    """Inverse of fold_tconv for gradients: returns (d conv_w [co,c_in,Kt,1], d conv_b,
    d align_w or None, d align_b or None).  When c_in <= c_out the align conv is unused
    and receives no gradient (the reference leaves .grad None)."""
    co = dW_eff.shape[1]
    dconv = np.zeros((co, c_in, Kt, 1), dtype=dW_eff.dtype)
    for k in range(Kt):
        dconv[:, :, k, 0] = dW_eff[k * c_in:(k + 1) * c_in, :].T
    if c_in > c_out:
        base = (Kt - 1) * c_in
        dalign_w = dW_eff[base:base + c_in, :c_out].T.reshape(c_out, c_in, 1, 1).copy()
        dalign_b = db_eff[:c_out].copy()
        return dconv, db_eff.copy(), dalign_w, dalign_b
    return dconv, db_eff.copy(), None, None


def fold_align(align_w, align_b, c_in, c_out):
    """Align before the graph conv (layers.py:223) as a dense map A = H @ Wa + ba, Wa [c_in, c_out]."""
    dt = align_w.dtype
    if c_in > c_out:
        return align_w[:, :, 0, 0].T.copy(), (np.zeros(c_out, dt) if align_b is None else align_b.copy())
    Wa = np.zeros((c_in, c_out), dtype=dt)
    for i in range(c_in):
        Wa[i, i] = 1.0
    return Wa, np.zeros(c_out, dtype=dt)


# ----------------------------------------------------------------------------- temporal conv
def im2col(x, Kt):
    """x (B,T,N,C) -> (B,T1,N,Kt*C), column index = tap*C + c  (valid conv along T, layers.py:55)."""
    B, T, N, C = x.shape
    T1 = T - Kt + 1
    return np.concatenate([x[:, k:k + T1] for k in range(Kt)], axis=-1)


def sigmoid(z):
    return 1.0 / (1.0 + np.exp(-z))


def tconv_fwd(x, W_eff, b_eff, Kt, c_out, act="glu", q=None):
    """Z = im2col(x) @ W_eff + b_eff; U = Z[..., :c_out] (= P + R), S = sigmoid(Z[..., c_out:]);
    GLU: H = U*S (layers.py:105);  GTU: H = tanh(U)*S (layers.py:109).  Returned U, S, H are unrounded (the caller stores q(U), q(S))."""
    Z = _q(q, im2col(x, Kt)) @ _q(q, W_eff) + b_eff
    U = Z[..., :c_out]
    S = sigmoid(Z[..., c_out:])
    H = U * S if act == "glu" else np.tanh(U) * S
    return U, S, H


def gate_bwd(dH, U, S, act="glu"):
    """dZ = [dU | dQ]  (SURVEY.md section 8a row a2)."""
    if act == "glu":
        dU = dH * S
        dQ = dH * U * S * (1.0 - S)
    else:
        th = np.tanh(U)
        dU = dH * S * (1.0 - th * th)
        dQ = dH * th * S * (1.0 - S)
    return np.concatenate([dU, dQ], axis=-1)


def tconv_bwd_data(dZ, W_eff, Kt, c_in, q=None):
    """dx[b,t,n,i] = sum_k sum_o dZ[b,t-k,n,o] * W_eff[k*c_in+i, o]   (0 <= t-k < T1)."""
    B, T1, N, co = dZ.shape
    T = T1 + Kt - 1
    dx = np.zeros((B, T, N, c_in), dtype=dZ.dtype)
    dZq, Wq = _q(q, dZ), _q(q, W_eff)
    for k in range(Kt):
        dx[:, k:k + T1] += dZq @ Wq[k * c_in:(k + 1) * c_in, :].T
    return dx


def tconv_bwd_weight(x, dZ, Kt, q=None):
    """dW_eff[k*c_in+i, o] = sum_{b,t,n} x[b,t+k,n,i] dZ[b,t,n,o];  db_eff = sum dZ (of the unrounded dZ)."""
    cols = _q(q, im2col(x, Kt))
    K = cols.shape[-1]
    dW = cols.reshape(-1, K).T @ _q(q, dZ).reshape(-1, dZ.shape[-1])
    db = dZ.reshape(-1, dZ.shape[-1]).sum(0)
    return dW, db


# ----------------------------------------------------------------------------- graph conv
def pack_gc_weight(weight, graph_conv_type):
    """Chebyshev weight (Ks,c,c) as is; Kipf weight (c,c) becomes a 2-term stack [0, W]
    (Y = (L X) W, layers.py:198-199)."""
    if graph_conv_type == "cheb_graph_conv":
        return weight
    return np.stack([np.zeros_like(weight), weight], axis=0)


def _gso_apply(L, X):
    """einsum('hi,btic->bthc', L, X) (layers.py:154/158/161) as ONE BLAS GEMM (N x N) . (N x B*T*C): the same sums in
    another order, minutes faster on the 8192-node graph."""
    B, T, N, C = X.shape
    return (L @ X.transpose(2, 0, 1, 3).reshape(N, B * T * C)).reshape(N, B, T, C).transpose(1, 2, 0, 3)


def cheb_polys(gso, Ks):
    """T_0 = I, T_1 = L, T_k = 2 L T_{k-1} - T_{k-2} of the operator itself (what stgcn_gso_prepare forms once per model for the
    slab-resident graph conv: float64 accumulation, stored as fp32)."""
    g = np.asarray(gso, dtype=np.float64)
    T = [np.eye(g.shape[0]), g]
    for k in range(2, Ks):
        T.append(2.0 * g @ T[k - 1] - T[k - 2])
    return [t.astype(np.float32).astype(np.float64) for t in T[:max(Ks, 1)]]


def gconv_fwd(A, gso, Wk, bias, q=None, form="recursion"):
    """X0=A, X1=L X0, Xk = 2 L X_{k-1} - X_{k-2} (layers.py:147-161); Y = sum_k Xk Wk + b (:165-168);
    G = relu(Y + A) (layers.py:229, 253).  Returns ([X0..X_{Ks-1}], G).
    With a rounding rule ``q`` the two ways the HIP path evaluates the terms differ and ``form`` selects one:
      "poly"      (graphs up to 512 nodes): X_k = q(T_k(L)) q(X0) with the precomputed polynomials, every term stored as q(X_k);
      "recursion" (tiled path): X_1 = q(L) q(X0), X_k = 2 q(L) q(X_{k-1}) - X_{k-2} on the stored terms, stored as q(X_k).
    A is the stored (already rounded) X0; the residual adds that stored value."""
    Ks = Wk.shape[0]
    Xs = [A]
    if q is not None and form == "poly":
        Tk = cheb_polys(gso, Ks)
        for k in range(1, Ks):
            Xs.append(q(_gso_apply(q(Tk[k]), q(A))))
    else:
        Lq = _q(q, gso)
        if Ks >= 2:
            Xs.append(_q(q, _gso_apply(Lq, _q(q, A))))
        for k in range(2, Ks):
            Xs.append(_q(q, 2.0 * _gso_apply(Lq, _q(q, Xs[k - 1])) - Xs[k - 2]))
    Y = sum(_q(q, Xs[k]) @ _q(q, Wk[k]) for k in range(Ks))
    if bias is not None:
        Y = Y + bias
    G = np.maximum(Y + A, 0.0)
    return Xs, G


def gconv_bwd(dG, G, Xs, gso, Wk, q=None, form="recursion"):
    """Backward of gconv_fwd: relu mask, then
    G_k = dY Wk^T; for k=Ks-1..2: G_{k-1} += 2 L^T G_k, G_{k-2} -= G_k; dA = G_0 + L^T G_1 + dY.
    With ``q``: dY is the STORED masked gradient (the caller passes q(dYg) as dG with G > 0 already applied or not -- the mask is
    idempotent); form "poly": dA = G_0 + sum_k q(T_k^T) q(G_k) + dY with G_k = q(dY) q(W_k^T) kept in fp32 between the products."""
    Ks = Wk.shape[0]
    dY = dG * (G > 0)
    if q is not None:
        dYq = q(dY)
        dWk = np.stack([q(Xs[k]).reshape(-1, Xs[k].shape[-1]).T @ dYq.reshape(-1, dY.shape[-1]) for k in range(Ks)])
        dbias = dYq.reshape(-1, dY.shape[-1]).sum(0)
        Gk = [dYq @ q(Wk[k]).T for k in range(Ks)]
        if form == "poly":
            Tk = cheb_polys(gso, Ks)
            dA = Gk[0] + dY
            for k in range(1, Ks):
                dA = dA + _gso_apply(q(Tk[k]).T, q(Gk[k]))
            return dA, dWk, dbias
        # tiled path: g_k = q(dY) q(W_k^T) (+ dY on k = 0) stored as q(g_k); then the Clenshaw recurrence in place over the stored
        # buffers, every operator product with q(L^T) and the stored operand, every result stored:
        #     b_K = g_K ; b_k = q(g_k + 2 L^T b_{k+1} - b_{k+2}) ; dA = g_0 + L^T b_1 - b_2
        K = Ks - 1
        g = [q(Gk[k] + (dY if k == 0 else 0.0)) for k in range(Ks)]
        if K == 0:
            return g[0], dWk, dbias
        LT = q(np.asarray(gso, dtype=np.float64)).T
        b = list(g)
        for k in range(K - 1, 0, -1):
            b[k] = q(g[k] + 2.0 * _gso_apply(LT, b[k + 1]) - (b[k + 2] if k + 2 <= K else 0.0))
        dA = g[0] + _gso_apply(LT, b[1]) - (b[2] if K >= 2 else 0.0)
        return dA, dWk, dbias
    dWk = np.stack([Xs[k].reshape(-1, Xs[k].shape[-1]).T @ dY.reshape(-1, dY.shape[-1]) for k in range(Ks)])
    dbias = dY.reshape(-1, dY.shape[-1]).sum(0)
    Gk = [dY @ Wk[k].T for k in range(Ks)]
    gT = gso.T
    for k in range(Ks - 1, 1, -1):
        Gk[k - 1] = Gk[k - 1] + 2.0 * _gso_apply(gT, Gk[k])
        Gk[k - 2] = Gk[k - 2] - Gk[k]
    dA = Gk[0] + dY
    if Ks >= 2:
        dA = dA + _gso_apply(gT, Gk[1])
    return dA, dWk, dbias


# ----------------------------------------------------------------------------- LayerNorm + dropout
def ln_dropout_fwd(H, gamma, beta, keep, p, eps=1e-12, H_norm=None):
    """LayerNorm over the joint [N, C] axes per (b, t), biased variance (layers.py:246, 255),
    then inverted dropout with an explicit keep mask (layers.py:256).  keep None -> eval.
    H_norm (bf16 statement only): the values that are normalised when they differ from the values the statistics were formed from --
    the kernels that run LayerNorm as a separate pass (ln_norm_kernel, the head's fc staging) rebuild H from the STORED gate inputs,
    while the statistics always come from the unrounded H of the conv epilogue."""
    mean = H.mean(axis=(2, 3), keepdims=True)
    var = ((H - mean) ** 2).mean(axis=(2, 3), keepdims=True)
    rstd = 1.0 / np.sqrt(var + eps)
    y = ((H if H_norm is None else H_norm) - mean) * rstd * gamma + beta
    if keep is not None:
        y = y * keep * (1.0 / (1.0 - p))
    return y, mean[..., 0, 0], rstd[..., 0, 0]


def ln_dropout_bwd(dy, H, gamma, mean, rstd, keep, p, y_stored=None, beta=None):
    """dH = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy_m * gamma;
    dgamma = sum_{b,t} dy_m * xhat; dbeta = sum_{b,t} dy_m.
    y_stored / beta (bf16 statement): the slab constant mean(g * xhat) is formed the way the HIP path forms it, from the block's STORED
    output y = mask * (xhat * gamma + beta) instead of from xhat:  sum g xhat = sum_kept dy (y - beta / (1 - p))  (stgcn_ln_hook);
    exact for the unrounded y, and the one place where the rounding of y enters the backward."""
    dym = dy if keep is None else dy * keep * (1.0 / (1.0 - p))
    xhat = (H - mean[..., None, None]) * rstd[..., None, None]
    g = dym * gamma
    c1 = g.mean(axis=(2, 3), keepdims=True)
    if y_stored is None:
        c2 = (g * xhat).mean(axis=(2, 3), keepdims=True)
    else:
        ks = 1.0 if keep is None else 1.0 / (1.0 - p)
        kept = 1.0 if keep is None else keep
        c2 = (dy * kept * (y_stored - ks * beta)).mean(axis=(2, 3), keepdims=True)
    dH = rstd[..., None, None] * (g - c1 - xhat * c2)
    return dH, (dym * xhat).sum(axis=(0, 1)), dym.sum(axis=(0, 1))


# ----------------------------------------------------------------------------- whole block
def block_params_np(p, prefix, graph_conv_type, dtype):
    g = lambda k: (p[prefix + k].detach().cpu().numpy().astype(dtype) if (prefix + k) in p else None)
    gc = "graph_conv.cheb_graph_conv." if graph_conv_type == "cheb_graph_conv" else "graph_conv.graph_conv."
    return dict(tc1_w=g("tmp_conv1.causal_conv.weight"), tc1_b=g("tmp_conv1.causal_conv.bias"),
                tc1_aw=g("tmp_conv1.align.align_conv.weight"), tc1_ab=g("tmp_conv1.align.align_conv.bias"),
                al_w=g("graph_conv.align.align_conv.weight"), al_b=g("graph_conv.align.align_conv.bias"),
                gc_w=g(gc + "weight"), gc_b=g(gc + "bias"),
                tc2_w=g("tmp_conv2.causal_conv.weight"), tc2_b=g("tmp_conv2.causal_conv.bias"),
                tc2_aw=g("tmp_conv2.align.align_conv.weight"), tc2_ab=g("tmp_conv2.align.align_conv.bias"),
                ln_w=g("tc2_ln.weight"), ln_b=g("tc2_ln.bias"))


def _gate(U, S, act):
    return U * S if act == "glu" else np.tanh(U) * S


def stblock_fwd(x, gso, bp, Kt, c_in, channels, graph_conv_type="cheb_graph_conv", act="glu",
                keep=None, p_drop=0.0, q=None, gc_form="poly", ln_from_stored=False):
    """x channels-last (B,T,N,c_in).  Returns (y (B,T2,N,c2), saved dict).
    ``q``: rounding rule of the bf16 configurations (QuantBf16; x must then hold bf16 values): `saved` carries what the HIP path
    stores (q(U1), q(S1), q(A), q(X_k), q(G)) plus the unrounded H1 / U2 / S2 / H2 of the forward (the backward recomputes the gate inputs
    of tmp_conv2 from G); y is q(dropout(LN(H2))).
    ln_from_stored: the block's LayerNorm runs as a separate pass over the stored gate inputs (more than 448 nodes: ln_norm_kernel)
    instead of inside tc2_ln_fwd_kernel, which normalises the values it still holds in registers."""
    c0, c1, c2 = channels
    W1, b1 = fold_tconv(bp["tc1_w"], bp["tc1_b"], bp["tc1_aw"], bp["tc1_ab"], c_in, c0, Kt)
    U1, S1, H1 = tconv_fwd(x, W1, b1, Kt, c0, act, q)
    Wa, ba = fold_align(bp["al_w"], bp["al_b"], c0, c1)
    A = _q(q, _q(q, H1) @ _q(q, Wa) + ba)
    Wk = pack_gc_weight(bp["gc_w"], graph_conv_type)
    Xs, G = gconv_fwd(A, gso, Wk, bp["gc_b"], q, gc_form)
    G = _q(q, G)
    W2, b2 = fold_tconv(bp["tc2_w"], bp["tc2_b"], bp["tc2_aw"], bp["tc2_ab"], c1, c2, Kt)
    U2, S2, H2 = tconv_fwd(G, W2, b2, Kt, c2, act, q)
    Hn = _gate(q(U2), q(S2), act) if (q is not None and ln_from_stored) else None
    y, mean, rstd = ln_dropout_fwd(H2, bp["ln_w"], bp["ln_b"], keep, p_drop, H_norm=Hn)
    y = _q(q, y)
    if q is not None:
        # backward of a bf16 block works on the STORED gate inputs -- except the cheap first conv (Kt*c_in <= 16), whose U1 / S1 are
        # not stored but recomputed from x with the rounded weights (stgcn_stblock_plan.recompute_tc1): those stay unrounded
        if Kt * c_in > 16:
            U1, S1 = q(U1), q(S1)
            H1 = _gate(U1, S1, act)
        # (U2 / S2 are not stored: tc2_bwd_kernel recomputes them from the stored G with the same rounded operands -> unrounded values)
    saved = dict(x=x, W1=W1, U1=U1, S1=S1, H1=H1, Wa=Wa, Xs=Xs, G=G, Wk=Wk, W2=W2, U2=U2, S2=S2, H2=H2,
                 mean=mean, rstd=rstd, keep=keep, A=A, y=y)
    return y, saved


def stblock_bwd(dy, sv, gso, bp, Kt, c_in, channels, graph_conv_type="cheb_graph_conv", act="glu",
                p_drop=0.0, need_dx=True, q=None, gc_form="poly", stages=None):
    """Returns (dx or None, grads dict keyed like block_params_np; unused tensors -> None).
    ``q``: see stblock_fwd (dy must hold bf16 values; `sv` must come from stblock_fwd with the same q).  ``stages``: optional dict that
    receives the stored intermediate gradients (dYg, dA) for stage-level comparisons."""
    c0, c1, c2 = channels
    H2 = sv["H2"] if q is None else _gate(sv["U2"], sv["S2"], act)      # bf16: xhat is rebuilt from the stored gate inputs
    dH2, dgamma, dbeta = ln_dropout_bwd(dy, H2, bp["ln_w"], sv["mean"], sv["rstd"], sv["keep"], p_drop,
                                        y_stored=None if q is None else sv["y"], beta=bp["ln_b"])
    dZ2 = gate_bwd(dH2, sv["U2"], sv["S2"], act)
    dW2, db2 = tconv_bwd_weight(sv["G"], dZ2, Kt, q)
    dG = tconv_bwd_data(dZ2, sv["W2"], Kt, c1, q)
    if q is not None:
        dG = q(dG * (sv["G"] > 0))                                       # stored dYg
    dA, dWk, dgb = gconv_bwd(dG, sv["G"], sv["Xs"], gso, sv["Wk"], q, gc_form)
    dA = _q(q, dA)
    if stages is not None:
        stages["dYg"], stages["dA"] = dG * (sv["G"] > 0), dA
    dH1 = _q(q, dA) @ _q(q, sv["Wa"]).T
    dWa = _q(q, sv["H1"]).reshape(-1, c0).T @ _q(q, dA).reshape(-1, c1)
    dba = dA.reshape(-1, c1).sum(0)
    dZ1 = gate_bwd(dH1, sv["U1"], sv["S1"], act)
    dW1, db1 = tconv_bwd_weight(sv["x"], dZ1, Kt, q)
    dx = _q(q, tconv_bwd_data(dZ1, sv["W1"], Kt, c_in, q)) if need_dx else None
    g = {}
    g["tc1_w"], g["tc1_b"], g["tc1_aw"], g["tc1_ab"] = unfold_tconv_grads(dW1, db1, c_in, c0, Kt)
    g["tc2_w"], g["tc2_b"], g["tc2_aw"], g["tc2_ab"] = unfold_tconv_grads(dW2, db2, c1, c2, Kt)
    if c0 > c1:
        g["al_w"], g["al_b"] = dWa.T.reshape(c1, c0, 1, 1).copy(), dba
    else:
        g["al_w"], g["al_b"] = None, None
    g["gc_w"] = dWk if graph_conv_type == "cheb_graph_conv" else dWk[1]
    g["gc_b"] = dgb
    g["ln_w"], g["ln_b"] = dgamma, dbeta
    return dx, g


# ----------------------------------------------------------------------------- output head (layers.py:260-284), explicit backward
def head_params_np(p, dtype, prefix="output."):
    g = lambda k: (p[prefix + k].detach().cpu().numpy().astype(dtype) if (prefix + k) in p else None)
    return dict(tc_w=g("tmp_conv1.causal_conv.weight"), tc_b=g("tmp_conv1.causal_conv.bias"),
                tc_aw=g("tmp_conv1.align.align_conv.weight"), tc_ab=g("tmp_conv1.align.align_conv.bias"),
                ln_w=g("tc1_ln.weight"), ln_b=g("tc1_ln.bias"), fc1_w=g("fc1.weight"), fc1_b=g("fc1.bias"),
                fc2_w=g("fc2.weight"), fc2_b=g("fc2.bias"))


def outblock_fwd(x, hp, Ko, c_in, channels, act="glu", keep=None, p_drop=0.0, q=None):
    """OutputBlock.forward (layers.py:276-284) on channels-last x (B, T, N, c_in): gated Ko-tap conv -> LayerNorm([N, c0]) -> fc1 -> ReLU
    -> dropout (explicit keep mask over (B, T1, N, c1), None = eval) -> fc2.  Returns (out (B, T1, N), saved dict).
    ``q`` (QuantBf16): stored tensors q(U), q(S), q(yln), q(hd); every matrix-product operand rounded; the prediction stays fp32 and is
    formed from the unrounded hidden values (a VALU dot product in the HIP path)."""
    c0, c1 = channels
    W, b = fold_tconv(hp["tc_w"], hp["tc_b"], hp["tc_aw"], hp["tc_ab"], c_in, c0, Ko)
    U, S, H = tconv_fwd(x, W, b, Ko, c0, act, q)
    Hn = None if q is None else _gate(q(U), q(S), act)      # the head's LayerNorm runs in the fc kernel's staging, on the stored gate inputs
    yln, mean, rstd = ln_dropout_fwd(H, hp["ln_w"], hp["ln_b"], None, 0.0, H_norm=Hn)
    W1 = hp["fc1_w"]                                                   # (c1, c0)
    h1 = _q(q, yln) @ _q(q, W1).T
    if hp["fc1_b"] is not None:
        h1 = h1 + hp["fc1_b"]
    h1 = np.maximum(h1, 0.0)
    hd = h1 if keep is None else h1 * keep * (1.0 / (1.0 - p_drop))
    out = hd @ hp["fc2_w"][0]
    if hp["fc2_b"] is not None:
        out = out + hp["fc2_b"][0]
    if q is not None:
        U, S = q(U), q(S)
    sv = dict(x=x, W=W, U=U, S=S, H=H, mean=mean, rstd=rstd, yln=_q(q, yln), hd=_q(q, hd), keep=keep)
    return out, sv


def outblock_bwd(dout, sv, hp, Ko, c_in, channels, act="glu", p_drop=0.0, need_dx=True, q=None, stages=None):
    """Backward of outblock_fwd; dout (B, T1, N) fp32.  Returns (dx or None, grads keyed like head_params_np)."""
    c0, c1 = channels
    hd, yln = sv["hd"], sv["yln"]
    scale = 1.0 if sv["keep"] is None else 1.0 / (1.0 - p_drop)
    go = dout[..., None]
    dh1 = np.where(hd != 0.0, go * hp["fc2_w"][0] * scale, 0.0)       # relu' and the dropout mask in one test (hd is zero where either is)
    dw2 = (go * hd).reshape(-1, c1).sum(0)[None, :]
    db2 = np.array([dout.sum()])
    dh1s = _q(q, dh1)                                                  # stored
    dyln = _q(q, _q(q, dh1) @ _q(q, hp["fc1_w"]))                      # (.., c0), stored
    dW1 = dh1s.reshape(-1, c1).T @ _q(q, yln).reshape(-1, c0)
    db1 = dh1s.reshape(-1, c1).sum(0)
    H = sv["H"] if q is None else _gate(sv["U"], sv["S"], act)
    dH, dgamma, dbeta = ln_dropout_bwd(dyln, H, hp["ln_w"], sv["mean"], sv["rstd"], None, 0.0)
    dZ = _q(q, gate_bwd(dH, sv["U"], sv["S"], act))                    # stored (the conv weight gradient reads it back)
    dW, db = tconv_bwd_weight(sv["x"], dZ, Ko, q)
    dx = _q(q, tconv_bwd_data(dZ, sv["W"], Ko, c_in, q)) if need_dx else None
    if stages is not None:
        stages.update(dh1=dh1s, dyln=dyln, dZ=dZ)
    g = {}
    g["tc_w"], g["tc_b"], g["tc_aw"], g["tc_ab"] = unfold_tconv_grads(dW, db, c_in, c0, Ko)
    g["ln_w"], g["ln_b"] = dgamma, dbeta
    g["fc1_w"], g["fc1_b"] = dW1, (db1 if hp["fc1_b"] is not None else None)
    g["fc2_w"], g["fc2_b"] = dw2, (db2 if hp["fc2_b"] is not None else None)
    return dx, g
