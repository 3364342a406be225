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


def tconv_fwd(x, W_eff, b_eff, Kt, c_out, act="glu"):
    """Z = im2col(x) @ W_eff + b_eff; U = Z[..., :c_out] (= P + R), S = sigmoid(Z[..., c_out:]);
    GLU: H = U*S (layers.py:105);  GTU: H = tanh(U)*S (layers.py:109)."""
    Z = im2col(x, Kt) @ W_eff + b_eff
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


def tconv_bwd_data(dZ, W_eff, Kt, c_in):
    """dx[b,t,n,i] = sum_k sum_o dZ[b,t-k,n,o] * W_eff[k*c_in+i, o]   (0 <= t-k < T1)."""
    B, T1, N, co = dZ.shape
    T = T1 + Kt - 1
    dx = np.zeros((B, T, N, c_in), dtype=dZ.dtype)
    for k in range(Kt):
        dx[:, k:k + T1] += dZ @ W_eff[k * c_in:(k + 1) * c_in, :].T
    return dx


def tconv_bwd_weight(x, dZ, Kt):
    """dW_eff[k*c_in+i, o] = sum_{b,t,n} x[b,t+k,n,i] dZ[b,t,n,o];  db_eff = sum dZ."""
    cols = im2col(x, Kt)
    K = cols.shape[-1]
    dW = cols.reshape(-1, K).T @ dZ.reshape(-1, dZ.shape[-1])
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


def gconv_fwd(A, gso, Wk, bias):
    """X0=A, X1=L X0, Xk = 2 L X_{k-1} - X_{k-2} (layers.py:147-161); Y = sum_k Xk Wk + b (:165-168);
    G = relu(Y + A) (layers.py:229, 253).  Returns ([X0..X_{Ks-1}], G)."""
    Ks = Wk.shape[0]
    Xs = [A]
    if Ks >= 2:
        Xs.append(_gso_apply(gso, A))
    for k in range(2, Ks):
        Xs.append(2.0 * _gso_apply(gso, Xs[k - 1]) - Xs[k - 2])
    Y = sum(Xs[k] @ Wk[k] for k in range(Ks))
    if bias is not None:
        Y = Y + bias
    G = np.maximum(Y + A, 0.0)
    return Xs, G


def gconv_bwd(dG, G, Xs, gso, Wk):
    """Backward of gconv_fwd: relu mask, then
    G_k = dY Wk^T; for k=Ks-1..2: G_{k-1} += 2 L^T G_k, G_{k-2} -= G_k; dA = G_0 + L^T G_1 + dY."""
    Ks = Wk.shape[0]
    dY = dG * (G > 0)
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
def ln_dropout_fwd(H, gamma, beta, keep, p, eps=1e-12):
    """LayerNorm over the joint [N, C] axes per (b, t), biased variance (layers.py:246, 255),
    then inverted dropout with an explicit keep mask (layers.py:256).  keep None -> eval."""
    mean = H.mean(axis=(2, 3), keepdims=True)
    var = ((H - mean) ** 2).mean(axis=(2, 3), keepdims=True)
    rstd = 1.0 / np.sqrt(var + eps)
    y = (H - mean) * rstd * gamma + beta
    if keep is not None:
        y = y * keep * (1.0 / (1.0 - p))
    return y, mean[..., 0, 0], rstd[..., 0, 0]


def ln_dropout_bwd(dy, H, gamma, mean, rstd, keep, p):
    """dH = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy_m * gamma;
    dgamma = sum_{b,t} dy_m * xhat; dbeta = sum_{b,t} dy_m."""
    dym = dy if keep is None else dy * keep * (1.0 / (1.0 - p))
    xhat = (H - mean[..., None, None]) * rstd[..., None, None]
    g = dym * gamma
    c1 = g.mean(axis=(2, 3), keepdims=True)
    c2 = (g * xhat).mean(axis=(2, 3), keepdims=True)
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


def stblock_fwd(x, gso, bp, Kt, c_in, channels, graph_conv_type="cheb_graph_conv", act="glu",
                keep=None, p_drop=0.0):
    """x channels-last (B,T,N,c_in).  Returns (y (B,T2,N,c2), saved dict)."""
    c0, c1, c2 = channels
    W1, b1 = fold_tconv(bp["tc1_w"], bp["tc1_b"], bp["tc1_aw"], bp["tc1_ab"], c_in, c0, Kt)
    U1, S1, H1 = tconv_fwd(x, W1, b1, Kt, c0, act)
    Wa, ba = fold_align(bp["al_w"], bp["al_b"], c0, c1)
    A = H1 @ Wa + ba
    Wk = pack_gc_weight(bp["gc_w"], graph_conv_type)
    Xs, G = gconv_fwd(A, gso, Wk, bp["gc_b"])
    W2, b2 = fold_tconv(bp["tc2_w"], bp["tc2_b"], bp["tc2_aw"], bp["tc2_ab"], c1, c2, Kt)
    U2, S2, H2 = tconv_fwd(G, W2, b2, Kt, c2, act)
    y, mean, rstd = ln_dropout_fwd(H2, bp["ln_w"], bp["ln_b"], keep, p_drop)
    saved = dict(x=x, W1=W1, U1=U1, S1=S1, H1=H1, Wa=Wa, Xs=Xs, G=G, Wk=Wk, W2=W2, U2=U2, S2=S2, H2=H2,
                 mean=mean, rstd=rstd, keep=keep, A=A)
    return y, saved


def stblock_bwd(dy, sv, gso, bp, Kt, c_in, channels, graph_conv_type="cheb_graph_conv", act="glu",
                p_drop=0.0, need_dx=True):
    """Returns (dx or None, grads dict keyed like block_params_np; unused tensors -> None)."""
    c0, c1, c2 = channels
    dH2, dgamma, dbeta = ln_dropout_bwd(dy, sv["H2"], bp["ln_w"], sv["mean"], sv["rstd"], sv["keep"], p_drop)
    dZ2 = gate_bwd(dH2, sv["U2"], sv["S2"], act)
    dW2, db2 = tconv_bwd_weight(sv["G"], dZ2, Kt)
    dG = tconv_bwd_data(dZ2, sv["W2"], Kt, c1)
    dA, dWk, dgb = gconv_bwd(dG, sv["G"], sv["Xs"], gso, sv["Wk"])
    dH1 = dA @ sv["Wa"].T
    dWa = sv["H1"].reshape(-1, c0).T @ dA.reshape(-1, c1)
    dba = dA.reshape(-1, c1).sum(0)
    dZ1 = gate_bwd(dH1, sv["U1"], sv["S1"], act)
    dW1, db1 = tconv_bwd_weight(sv["x"], dZ1, Kt)
    dx = tconv_bwd_data(dZ1, sv["W1"], Kt, c_in) if need_dx else None
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
