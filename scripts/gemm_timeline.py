"""Per-block cycle timeline of the persistent 256x256 GEMM kernel (s_memtime stamps; profiling hook)."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from capreolus_amd import _lib
lib = _lib.profiling_build().__enter__(); dev = "cuda:0"   # (the -DCAPAMD_PROFILING build for the whole script)
vp = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
M = 65536
for name, N, K, epi in [("qkv-like", 2304, 768, 0), ("ffn1", 3072, 768, 1), ("ffn1 out chunk-major", 3072, 768, 0x101), ("oproj", 768, 768, 0),
                        ("ffn2", 768, 3072, 0), ("ffn2 A chunk-major", 768, 3072, 0x200)]:
    A = torch.randn((M, K), device=dev).bfloat16(); W = (torch.randn((N, K), device=dev) * 0.05).bfloat16()
    bias = torch.randn(N, device=dev); resid = torch.randn((M, N), device=dev).bfloat16() if (epi & 0xff) == 4 else None
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    stamps = torch.zeros((256, 32), dtype=torch.int64, device=dev)
    for _ in range(2):
        lib.capamd_bert_gemm(vp(A), vp(W), vp(bias), M, N, K, epi, vp(resid), vp(out), 0, st)
    lib.capamd_debug_set_gemm_stamps(vp(stamps))
    lib.capamd_bert_gemm(vp(A), vp(W), vp(bias), M, N, K, epi, vp(resid), vp(out), 0, st)
    torch.cuda.synchronize()
    lib.capamd_debug_set_gemm_stamps(None)
    s = stamps.cpu().numpy()
    n = int((s[0] != 0).sum())
    d = (s[:, 1:n] - s[:, : n - 1]).astype("float64")
    med = np.median(d, axis=0)
    print(name, "stamps", n, "[start | (k-loop, epilogue) per tile] median cycle deltas:", [int(x) for x in med], "total", int(med.sum()))
