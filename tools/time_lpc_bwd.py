"""LPC branch with a gradient at the bench size (1024 utterances x 1 s, 204 800 frames): the module chain LPC(Window(Frame(x))),
its pieces, and -- when the library has it -- the one-launch forward / backward pair behind fuse(frame, window, lpc)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffsptk_amd as dsp
from diffsptk_amd import ops, _lib
dev = "cuda"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
x = torch.randn(B, 16000, device=dev)
frm, wn, lpc = dsp.Frame(400, 80), dsp.Window(400, device=dev), dsp.LPC(400, 24, eps=1e-5, device=dev)


def t(fn, n=20, rounds=5):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / n)
    return sorted(ts)[len(ts) // 2]


with torch.no_grad():
    xw = wn(frm(x)).contiguous()
    a = lpc(xw)
F = B * 200
g = torch.randn(F, 25, device=dev)
gx = torch.empty(F, 400, device=dev)
xw2, a2 = xw.view(F, 400), a.view(F, 25).contiguous()
print("dsa_lpc_bwd alone (framed rows in, framed cotangent out): %.4f ms" % t(lambda: ops._call("dsa_lpc_bwd", g.data_ptr(), xw2.data_ptr(), a2.data_ptr(), F, 400, 24, 1e-5, _lib.F32, gx.data_ptr(), ops._stream())), _lib.last_kernel())
with torch.no_grad():
    print("module chain forward: %.4f ms" % t(lambda: lpc(wn(frm(x)))))
    print("fused forward (ops.frame_window_lpc): %.4f ms" % t(lambda: ops.frame_window_lpc(x, wn.window, 400, 80, 24, 1e-5)))


def chain():
    xg = x.detach().requires_grad_(True)
    lpc(wn(frm(xg))).mean().backward()


print("module chain forward + backward: %.4f ms" % t(chain, n=10))
if hasattr(dsp.modules.fused, "FusedFrameWindowLPC"):
    fl = dsp.fuse(frm, wn, lpc)

    def fused_fb():
        xg = x.detach().requires_grad_(True)
        fl(xg).mean().backward()

    with torch.no_grad():
        print("fuse(frame, window, lpc) forward: %.4f ms" % t(lambda: fl(x)), fl.last_path)
    print("fuse(frame, window, lpc) forward + backward: %.4f ms" % t(fused_fb, n=10), _lib.last_kernel())
    gg = torch.randn(B, 200, 25, device=dev)
    xg = x.detach().requires_grad_(True)
    y = fl(xg)
    print("  backward launch alone: %.4f ms" % t(lambda: torch.autograd.grad(y, xg, gg, retain_graph=True)))
gxd = torch.empty_like(x)
gd = torch.randn(B, 200, 25, device=dev)
print("  dsa_frame_window_lpc_bwd, direct calls: %.4f ms" % t(lambda: ops._call("dsa_frame_window_lpc_bwd", gd.data_ptr(), x.data_ptr(), B, 16000, 400, 80, wn.window.data_ptr(), 1, 0, 24,
                                                                             1e-5, _lib.F32, gxd.data_ptr(), ops._stream()), n=30))
