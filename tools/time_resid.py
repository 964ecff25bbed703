"""Timing of the Newton-residual launch and of the plain row product over the batch size (tools/, not part of the product)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffsptk_amd import ops

dev = "cuda"
K, M1 = 1025, 50
N = 2 * M1 - 1
g = torch.Generator().manual_seed(0)
D = (torch.randn(M1, K, generator=g) / M1 ** 0.5).to(dev)
E = (torch.randn(K, N, generator=g) / K).to(dev)


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for F in ((3200, 204800) if os.environ.get('ABL_ONLY') else (3200, 6400, 12800, 16384, 25600, 51200, 102400, 204800)):
    logx = (torch.randn(F, K, generator=g) * 0.5).to(dev)
    mc = (torch.randn(F, M1, generator=g) * 0.05).to(dev)
    tr = t(lambda: ops.mcep_newton_resid(logx, mc, D, E))
    tg = t(lambda: ops.rows_gemm(logx, E))
    fl_r = 2.0 * F * K * (N + 13 + 2 * M1 + 6)   # useful flops of the residual (products only)
    fl_g = 2.0 * F * K * N
    print(f"F {F:7d}: resid {tr:8.1f} us ({2.0 * F * K * (112 + 32 * 14 / 8 * 0 + 56) / tr / 1e6:6.1f} TF issued)  rows_gemm {tg:8.1f} us ({2.0 * F * K * 112 / tg / 1e6:6.1f} TF issued)")
