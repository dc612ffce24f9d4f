"""Fill-rate probe: K-loop cycle stamps of the 256x256 GEMM on a grid of few / many CUs (profiling only).
Run it against the default library and against a -DCAPAMD_GEMM_ABLATE=1|2|3 build (CAPAMD_LIB_PATH, see scripts/gemm_variants.sh)."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from capreolus_amd import _lib
lib = _lib.profiling_build().__enter__(); dev = "cuda:0"   # (the -DCAPAMD_PROFILING build for the whole script)
vp = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for name, M, N, K in [("8 CUs", 2048, 256, 3072), ("32 CUs", 8192, 256, 3072), ("64 CUs", 16384, 256, 3072), ("256 CUs x1 tile", 65536, 256, 3072), ("256 CUs, N=768", 65536, 768, 3072)]:
    A = torch.randn((M, K), device=dev).bfloat16(); W = (torch.randn((N, K), device=dev) * 0.05).bfloat16()
    bias = torch.randn(N, device=dev); out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    stamps = torch.zeros((256, 32), dtype=torch.int64, device=dev)
    for _ in range(2):
        lib.capamd_bert_gemm(vp(A), vp(W), vp(bias), M, N, K, 0, None, vp(out), 0, st)
    lib.capamd_debug_set_gemm_stamps(vp(stamps))
    lib.capamd_bert_gemm(vp(A), vp(W), vp(bias), M, N, K, 0, None, vp(out), 0, st)
    torch.cuda.synchronize()
    lib.capamd_debug_set_gemm_stamps(None)
    s = stamps.cpu().numpy(); nb = int((s[:, 0] != 0).sum())
    d = (s[:nb, 1] - s[:nb, 0]).astype("float64")
    print(f"{name:18s} blocks {nb:3d}  first tile K loop: median {np.median(d)/(K/64):7.0f} cycles per K step  ({65536/(np.median(d)/(K/64)):.1f} B/clk/CU)")
