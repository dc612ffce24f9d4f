"""Per-block cycle timeline (s_memtime stamps) of the 256-row ring GEMM on 16x16x32 MFMAs next to the 32x32x16 kernels, all chunk-major."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from capreolus_amd import _lib
lib = _lib.profiling_build().__enter__(); dev = "cuda:0"
vp = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
M = 64000
def to_cm(x):
    M_, C = x.shape
    return x.reshape(M_ // 32, 32, C // 8, 8).permute(0, 2, 1, 3).contiguous().reshape(-1)
SHAPES = [("qkv-like bias", 2304, 768, 0), ("ffn1 gelu", 3072, 768, 1), ("oproj resid", 768, 768, 5), ("ffn2 resid", 768, 3072, 5)]
if os.environ.get("ONLY"):
    SHAPES = [x for x in SHAPES if x[0].startswith(os.environ["ONLY"])]
KERNELS = ((0x700, "ring128/32"), (0x1F00, "ring256/32"), (0xF00, "ring256/16"), (0x2700, "ring128/16"))
if os.environ.get("ONLY16"):
    KERNELS = KERNELS[2:]
for name, N, K, epi in SHAPES:
    A = to_cm((torch.randn((M, K), device=dev) * 0.5).bfloat16()); Wc = to_cm((torch.randn((N, K), device=dev) * 0.05).bfloat16())
    bias = torch.randn(N, device=dev)
    out = torch.empty(M * N, dtype=torch.bfloat16, device=dev)
    R = to_cm((torch.randn((M, N), device=dev)).bfloat16()) if epi == 5 else None
    mr = torch.stack([torch.zeros(M, device=dev), torch.ones(M, device=dev)], 1).contiguous()
    gamma = torch.ones(N, device=dev); part = torch.zeros((M, N // 64, 2), device=dev)
    def call(flags):
        if epi == 5:
            return lib.capamd_bert_gemm_ln(vp(A), vp(Wc), vp(bias), M, N, K, 5 | flags, None, None, None, None, vp(R), vp(mr), vp(gamma), vp(part), vp(out), 0, st)
        return lib.capamd_bert_gemm(vp(A), vp(Wc), vp(bias), M, N, K, epi | flags, None, vp(out), 0, st)
    for flags, tag in KERNELS:
        stamps = torch.zeros((512, 32), dtype=torch.int64, device=dev)
        ts = []
        for i in range(8):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            assert call(flags) == 0
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        lib.capamd_debug_set_gemm_stamps(vp(stamps))
        call(flags)
        torch.cuda.synchronize()
        lib.capamd_debug_set_gemm_stamps(None)
        s = stamps.cpu().numpy()
        n = int((s[0] != 0).sum())
        d = (s[:256, 1:n] - s[:256, : n - 1]).astype("float64")
        med = np.median(d, axis=0)
        t = sorted(ts)[len(ts) // 2]
        print(f"{name:14s} {tag:11s} {t:7.1f} us {2.0*M*N*K/t/1e6:7.1f} TF | stamps {n} median cycle deltas [setup | (k-loop, epilogue)...]:", [int(x) for x in med][:7], "total", int(med.sum()), f"-> {med.sum()/t/1e3:.2f} GHz")
