"""K-loop cycles per K step of the 256x256 GEMM as a function of problem size / cache residency (profiling only)."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from capreolus_amd import _lib
lib = _lib.profiling_build().__enter__(); dev = "cuda:0"   # (the -DCAPAMD_PROFILING build for the whole script)
vp = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for name, M, N, K in [("M=2048  N=2304 (72 tiles, A 3 MB)", 2048, 2304, 768), ("M=8192  N=2304 (288 tiles)", 8192, 2304, 768),
                      ("M=65536 N=2304", 65536, 2304, 768), ("M=65536 N=256 K=768", 65536, 256, 768), ("M=2048 N=256 K=3072 (8 tiles)", 2048, 256, 3072)]:
    A = torch.randn((M, K), device=dev).bfloat16(); W = (torch.randn((N, K), device=dev) * 0.05).bfloat16()
    bias = torch.randn(N, device=dev); out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    stamps = torch.zeros((256, 32), dtype=torch.int64, device=dev)
    for _ in range(3):
        lib.capamd_bert_gemm(vp(A), vp(W), vp(bias), M, N, K, 0, None, vp(out), 0, st)
    lib.capamd_debug_set_gemm_stamps(vp(stamps))
    lib.capamd_bert_gemm(vp(A), vp(W), vp(bias), M, N, K, 0, None, vp(out), 0, st)
    torch.cuda.synchronize()
    lib.capamd_debug_set_gemm_stamps(None)
    s = stamps.cpu().numpy(); nb = int((s[:, 0] != 0).sum())
    first = (s[:nb, 1] - s[:nb, 0]).astype("float64")
    line = f"{name:36s} blocks {nb:3d}  first tile: {np.median(first)/(K/64):6.0f} cyc/K-step"
    if s[0, 3] != 0:
        second = (s[:nb, 3] - s[:nb, 2]).astype("float64")
        line += f"   second tile: {np.median(second)/(K/64):6.0f}"
    print(line)
