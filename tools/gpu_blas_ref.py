"""Reference point for the bf16 operator GEMM of the tiled graph conv: what the vendor library (hipBLASLt / rocBLAS through torch.matmul)
reaches on the same shapes on this GPU.  Not part of the product path; numbers go to profiles/."""
import json
import sys
import torch

def run(m, n, k, iters=20):
    a = torch.randn(m, k, device="cuda", dtype=torch.bfloat16)
    b = torch.randn(k, n, device="cuda", dtype=torch.bfloat16)
    bt = torch.randn(n, k, device="cuda", dtype=torch.bfloat16)
    out = {}
    for name, f in (("nn", lambda: a @ b), ("nt", lambda: a @ bt.t())):
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            f()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / iters
        out[name] = {"us": round(us, 1), "TF": round(2.0 * m * n * k / us / 1e6, 1)}
    return out

res = {}
for (m, n, k) in ((8192, 2560, 8192), (8192, 1536, 8192), (8192, 8192, 8192)):
    res[f"{m}x{n}x{k}"] = run(m, n, k)
print(json.dumps(res))
