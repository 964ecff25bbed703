"""Minimal script for kernel traces of the MLSA filter modes (256 utterances x 1 s, order 24)."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import diffsptk_amd as dsp
dev = "cuda"
mode = os.environ.get("MODE", "single-stage")
x = torch.randn(256, 16000, device=dev)
mc = 0.1 * torch.randn(256, 200, 25, device=dev)
f = dsp.PseudoMGLSADigitalFilter(24, 80, alpha=0.42, mode=mode, device=dev)
with torch.no_grad():
    for _ in range(4):
        f(x, mc)
torch.cuda.synchronize()
