"""Times capamd_bert_gemm on the four BERT-base GEMM shapes (M = 256 passages x 256 tokens)."""
import ctypes
import sys

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from capreolus_amd import _lib  # noqa: E402

lib = _lib.load()
dev = "cuda:0"
vp = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
M = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
shapes = [("qkv-like N=2304 K=768 bias", 2304, 768, 0), ("oproj N=768 K=768 resid", 768, 768, 4), ("ffn1 N=3072 K=768 gelu", 3072, 768, 1),
          ("ffn2 N=768 K=3072 resid", 768, 3072, 4)]
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
tot_f, tot_t = 0.0, 0.0
for name, N, K, epi in shapes:
    A = torch.randn((M, K), device=dev).bfloat16()
    W = (torch.randn((N, K), device=dev) * 0.05).bfloat16()
    bias = torch.randn(N, device=dev)
    resid = torch.randn((M, N), device=dev) if epi == 2 else (torch.randn((M, N), device=dev).bfloat16() if epi == 4 else None)
    out = torch.empty((M, N), dtype=torch.float32 if epi == 2 else torch.bfloat16, device=dev)
    ts = []
    for i in range(13):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        assert lib.capamd_bert_gemm(vp(A), vp(W), vp(bias), M, N, K, epi, vp(resid), vp(out), 0, st) == 0
        e1.record()
        torch.cuda.synchronize()
        if i >= 3:
            ts.append(e0.elapsed_time(e1) * 1e-3)
    t = sorted(ts)[len(ts) // 2]
    fl = 2.0 * M * N * K
    tot_f += fl
    tot_t += t
    # the vendor library on the same shape (torch -> hipBLASLt), bias only: no GELU / residual / QKV re-layout work
    bb = bias.bfloat16()
    tv = []
    for i in range(13):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        torch.nn.functional.linear(A, W, bb)
        e1.record()
        torch.cuda.synchronize()
        if i >= 3:
            tv.append(e0.elapsed_time(e1) * 1e-3)
    v = sorted(tv)[len(tv) // 2]
    print(f"{name:32s} {t*1e6:8.1f} us  {fl/t/1e12:7.1f} TF/s   | hipBLASLt linear+bias {v*1e6:8.1f} us {fl/v/1e12:7.1f} TF/s")
print(f"{'all four':32s} {tot_t*1e6:8.1f} us  {tot_f/tot_t/1e12:7.1f} TF/s")
