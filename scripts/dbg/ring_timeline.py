"""Per-block cycle timeline of the ring GEMM vs the ping-pong GEMM on chunk-major operands (s_memtime stamps)."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from capreolus_amd import _lib
lib = _lib.profiling_build().__enter__(); dev = "cuda:0"   # (the -DCAPAMD_PROFILING build for the whole script)
vp = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
M = 65536
def to_cm(x):
    M_, C = x.shape
    return x.reshape(M_ // 32, 32, C // 8, 8).permute(0, 2, 1, 3).contiguous().reshape(-1)
for name, N, K, epi in [("ffn1 gelu", 3072, 768, 1), ("plain N=3072", 3072, 768, 0), ("oproj-like", 768, 768, 0), ("ffn2-like", 768, 3072, 0)]:
    A = to_cm((torch.randn((M, K), device=dev) * 0.5).half()); W0 = (torch.randn((N, K), device=dev) * 0.05).half(); Wc = to_cm(W0)
    bias = torch.randn(N, device=dev)
    out = torch.empty(M * N, dtype=torch.float16, device=dev)
    for flags, W, tag in ((0x300, W0, "pingpong"), (0x700, Wc, "ring128"), (0xF00, Wc, "ring256")):
        stamps = torch.zeros((256, 32), dtype=torch.int64, device=dev)
        ts = []
        for i in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            assert lib.capamd_bert_gemm(vp(A), vp(W), vp(bias), M, N, K, epi | flags, None, vp(out), 1, st) == 0
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        lib.capamd_debug_set_gemm_stamps(vp(stamps))
        lib.capamd_bert_gemm(vp(A), vp(W), vp(bias), M, N, K, epi | flags, None, vp(out), 1, st)
        torch.cuda.synchronize()
        lib.capamd_debug_set_gemm_stamps(None)
        s = stamps.cpu().numpy()
        n = int((s[0] != 0).sum())
        d = (s[:, 1:n] - s[:, : n - 1]).astype("float64")
        med = np.median(d, axis=0)
        t = sorted(ts)[len(ts) // 2]
        print(f"{name:14s} {tag:9s} {t:7.1f} us {2.0*M*N*K/t/1e6:7.1f} TF | stamps {n} median cycle deltas [setup | (k-loop, epilogue)...]:", [int(x) for x in med][:9], "total", int(med.sum()))
