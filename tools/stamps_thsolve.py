"""Phase durations of thsolve_quadn_kernel from a -DTQ_STAMPS build (tools/build_variant.sh build/libtq_stamps.so thsolve_quad.hip -DTQ_STAMPS,
copied over the library): counter ticks of staging / construction / elimination / back substitution of wave tiles."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffsptk_amd import ops

dev = "cuda"
for n in (50, 35, 25):
    F = int(os.environ.get('F', '12800'))
    g = torch.Generator().manual_seed(0)
    rt = torch.randn(F, 2 * n - 1, generator=g).to(dev) * 0.01
    rt[:, 0] += 4.0
    av = torch.zeros(n, device=dev)
    mc = torch.zeros(F, n, device=dev)
    out = ops.mcep_newton_update(rt, av, mc)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        out = ops.mcep_newton_update(rt, av, mc)
    b.record()
    torch.cuda.synchronize()
    o = out.cpu()
    st = o[:, [0, 4, 8, 12]]
    print(f"n {n}: {a.elapsed_time(b) * 100:.1f} us per launch; ticks (median over systems) staging {st[:, 0].median():.0f} construction {st[:, 1].median():.0f} "
          f"elimination {st[:, 2].median():.0f} back substitution {st[:, 3].median():.0f}; max total {st.sum(1).max():.0f}")
