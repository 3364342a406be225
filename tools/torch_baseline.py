"""The reference's model as plain PyTorch-ROCm ops (MIOpen conv2d, rocBLAS einsum/bmm, ATen LayerNorm / dropout / pointwise):
the "unfused, same GPU" baseline SURVEY.md section 8d asks to report beside the fused path.  It follows the arithmetic of
hazdzz/STGCN model/layers.py:87-120 (TemporalConvLayer), :143-172 (ChebGraphConv), :222-231 (GraphConvLayer), :250-258
(STConvBlock), :276-284 (OutputBlock) and model/models.py:28-53 with the reference's state_dict keys, so that it can be loaded
from a drop-in model's state_dict.  Used ONLY by bench.py's `gpu_baseline` leg and tools/: it is not part of the product and
not the parity oracle (oracle/ pins parity; this is a timing yardstick)."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class _Align(nn.Module):
    def __init__(self, c_in, c_out):
        super().__init__()
        self.c_in, self.c_out = c_in, c_out
        self.align_conv = nn.Conv2d(c_in, c_out, (1, 1))

    def forward(self, x):
        if self.c_in > self.c_out:
            return self.align_conv(x)
        if self.c_in < self.c_out:
            return torch.cat([x, torch.zeros(x.shape[0], self.c_out - self.c_in, x.shape[2], x.shape[3], device=x.device, dtype=x.dtype)], dim=1)
        return x


class _TConv(nn.Module):
    def __init__(self, Kt, c_in, c_out):
        super().__init__()
        self.Kt, self.c_out = Kt, c_out
        self.align = _Align(c_in, c_out)
        self.causal_conv = nn.Conv2d(c_in, 2 * c_out, (Kt, 1))

    def forward(self, x):
        x_in = self.align(x)[:, :, self.Kt - 1:, :]
        z = self.causal_conv(x)
        return (z[:, :self.c_out] + x_in) * torch.sigmoid(z[:, -self.c_out:])


class _Cheb(nn.Module):
    def __init__(self, c, Ks, gso):
        super().__init__()
        self.Ks, self.gso = Ks, gso
        self.weight = nn.Parameter(torch.empty(Ks, c, c))
        self.bias = nn.Parameter(torch.empty(c))

    def forward(self, x):
        x = x.permute(0, 2, 3, 1)
        xs = [x]
        if self.Ks > 1:
            xs.append(torch.einsum("hi,btij->bthj", self.gso, x))
        for k in range(2, self.Ks):
            xs.append(torch.einsum("hi,btij->bthj", 2 * self.gso, xs[k - 1]) - xs[k - 2])
        return torch.einsum("btkhi,kij->bthj", torch.stack(xs, dim=2), self.weight) + self.bias


class _GCLayer(nn.Module):
    def __init__(self, c_in, c_out, Ks, gso):
        super().__init__()
        self.align = _Align(c_in, c_out)
        self.cheb_graph_conv = _Cheb(c_out, Ks, gso)

    def forward(self, x):
        x_in = self.align(x)
        return self.cheb_graph_conv(x_in).permute(0, 3, 1, 2) + x_in


class _STBlock(nn.Module):
    def __init__(self, Kt, Ks, N, c_in, ch, gso, p):
        super().__init__()
        self.tmp_conv1 = _TConv(Kt, c_in, ch[0])
        self.graph_conv = _GCLayer(ch[0], ch[1], Ks, gso)
        self.tmp_conv2 = _TConv(Kt, ch[1], ch[2])
        self.tc2_ln = nn.LayerNorm([N, ch[2]], eps=1e-12)
        self.dropout = nn.Dropout(p)

    def forward(self, x):
        x = torch.relu(self.graph_conv(self.tmp_conv1(x)))
        x = self.tmp_conv2(x)
        return self.dropout(self.tc2_ln(x.permute(0, 2, 3, 1)).permute(0, 3, 1, 2))


class _Head(nn.Module):
    def __init__(self, Ko, c_in, ch, end, N, p):
        super().__init__()
        self.tmp_conv1 = _TConv(Ko, c_in, ch[0])
        self.fc1 = nn.Linear(ch[0], ch[1])
        self.fc2 = nn.Linear(ch[1], end)
        self.tc1_ln = nn.LayerNorm([N, ch[0]], eps=1e-12)
        self.dropout = nn.Dropout(p)

    def forward(self, x):
        x = self.tc1_ln(self.tmp_conv1(x).permute(0, 2, 3, 1))
        return self.fc2(self.dropout(torch.relu(self.fc1(x)))).permute(0, 3, 1, 2)


class TorchSTGCNCheb(nn.Module):
    """STGCNChebGraphConv (models.py:28-53) for the glu / cheb_graph_conv configuration of BASELINE.json configs[1]."""

    def __init__(self, Kt, Ks, n_his, blocks, N, gso, droprate):
        super().__init__()
        n_st = len(blocks) - 3
        self.st_blocks = nn.Sequential(*[_STBlock(Kt, Ks, N, blocks[l][-1], blocks[l + 1], gso, droprate) for l in range(n_st)])
        Ko = n_his - n_st * 2 * (Kt - 1)
        self.output = _Head(Ko, blocks[-3][-1], blocks[-2], blocks[-1][0], N, droprate)

    def forward(self, x):
        return self.output(self.st_blocks(x))


def time_train_step(state_dict, gso, x, y, Kt=3, Ks=3, n_his=12, blocks=None, droprate=0.5, steps=20, warmup=5):
    """ms per step of the reference loop body (main.py:165-169: zero_grad, forward, MSELoss, backward, AdamW) through stock ops."""
    dev = x.device
    m = TorchSTGCNCheb(Kt, Ks, n_his, blocks, x.shape[-1], gso, droprate).to(dev)
    m.load_state_dict(state_dict, strict=True)
    m.train()
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3, weight_decay=1e-3)
    loss_fn = nn.MSELoss()

    def step():
        opt.zero_grad()
        loss = loss_fn(m(x).view(len(x), -1), y)
        loss.backward()
        opt.step()
        return loss

    for _ in range(warmup):
        step()
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        loss = step()
    e1.record()
    torch.cuda.synchronize(dev)
    return e0.elapsed_time(e1) / steps, float(loss)
