"""What does ONE fork / join pair cost inside a replayed hipGraph on this stack, and do kernels of the two branches overlap?
main chain: K dependent medium kernels (each ~10 us on a quarter of the chip); side branch: one such kernel forked after kernel 2, joined before kernel K-1.
Prints the replay time per variant (us)."""
import time
import torch

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)


def body(x):   # ~a few us: a small GEMM-free elementwise chain on 64 workgroups' worth of data
    for _ in range(4):
        x.mul_(1.0001).add_(0.5)


def build(kind, K=12, n=1 << 16):
    xs = [torch.zeros(n, device=dev) for _ in range(2)]
    side = torch.cuda.Stream(device=dev)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream(device=dev)
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            body(xs[0]); body(xs[1])
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        main = torch.cuda.current_stream()
        for k in range(K):
            if k == 2 and kind == "fork":
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    body(xs[1])
            if k == 2 and kind == "serial":
                body(xs[1])
            if k == K - 1 and kind == "fork":
                main.wait_stream(side)
            body(xs[0])
    return g


for kind in ("none", "serial", "fork"):
    g = build(kind)
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    R = 300
    for _ in range(R):
        g.replay()
    torch.cuda.synchronize()
    print(f"{kind:7s}: {(time.perf_counter() - t0) / R * 1e6:8.1f} us per replay")
