"""The four BERT-base encoder GEMM shapes at the benchmark's micro-batch (M = 250 passages x 256 tokens) in ONE session:
the ring kernel (bert_gemm_ring.h; chunk-major operands, 128-row tiles x 2 workgroups per CU and 256-row tiles x 1), the 8-wave
ping-pong kernel, and the vendor library on the same shape (torch.nn.functional.linear -> hipBLASLt, bias only).  Median of 20 timed
launches after 5 warm-up launches, HIP events around each launch.  `profiles/r03/gemm_bench_vs_hipblaslt.txt` is this script's output."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from capreolus_amd import _lib  # noqa: E402

lib = _lib.load()
dev = "cuda:0"
vp = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
M = int(sys.argv[1]) if len(sys.argv) > 1 else 64000
dt_name = sys.argv[2] if len(sys.argv) > 2 else "bf16"
tdt, code = (torch.bfloat16, 0) if dt_name == "bf16" else (torch.float16, 1)      # (capamd_bert_gemm: dtype 0 = bf16, 1 = fp16; rounds 3-4 had them swapped here)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def to_cm(t):
    R, C = t.shape
    return t.view(R // 32, 32, C // 8, 8).permute(0, 2, 1, 3).contiguous().view(-1)


def med(fn, n=20, warm=5):
    ts = []
    for i in range(n + warm):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        if i >= warm:
            ts.append(e0.elapsed_time(e1) * 1e-3)
    return sorted(ts)[len(ts) // 2]


shapes = [("QKV   N=2304 K=768  bias", 2304, 768, 0), ("Oproj N=768  K=768  bias", 768, 768, 0), ("FFN1  N=3072 K=768  bias+GELU", 3072, 768, 1),
          ("FFN2  N=768  K=3072 bias", 768, 3072, 0)]
if os.environ.get("GEMM_FFN1_BIAS_ONLY"):      # what the GELU epilogue costs: the FFN1 shape once more with the bias-only epilogue
    shapes.append(("FFN1* N=3072 K=768  bias only", 3072, 768, 0))
print(f"M = {M}, {dt_name} operands, fp32 accumulate; us per launch (TFLOP/s)")
print(f"{'shape':30s} {'ring 128 rows 32x32x16':>24s} {'ring 256 rows 32x32x16':>24s} {'ring 256 rows 16x16x32':>24s} {'ring 128 rows 16x16x32':>24s} {'ping-pong 256x256':>22s} {'hipBLASLt (bias only)':>24s}   best in-tree / vendor")
tot = {"ring128": 0.0, "ring256": 0.0, "ring256_16": 0.0, "ring128_16": 0.0, "pp": 0.0, "vendor": 0.0, "best": 0.0}
only = os.environ.get("GEMM_ONLY")            # e.g. GEMM_ONLY=FFN1: one shape (sweeps of CAPAMD_GEMM_NGROUP / CAPAMD_RING_* builds)
for name, N, K, epi in shapes:
    if only and not name.startswith(only):
        continue
    g = torch.Generator(device=dev)
    g.manual_seed(N + K)
    A = torch.randn((M, K), generator=g, device=dev).to(tdt)
    W = (torch.randn((N, K), generator=g, device=dev) * 0.05).to(tdt)
    bias = torch.randn(N, generator=g, device=dev)
    A_cm, W_cm = to_cm(A), to_cm(W)
    out = torch.empty(M * N, dtype=tdt, device=dev)

    def run(flags, a, w):
        assert lib.capamd_bert_gemm(vp(a), vp(w), vp(bias), M, N, K, epi | flags, None, vp(out), code, st) == 0

    t128 = med(lambda: run(0x700, A_cm, W_cm))
    t256 = med(lambda: run(0x1F00, A_cm, W_cm))
    t256_16 = med(lambda: run(0xF00, A_cm, W_cm))
    t128_16 = med(lambda: run(0x2700, A_cm, W_cm))
    tpp = med(lambda: run(0x300, A_cm, W))
    bb = bias.to(tdt)
    tv = med(lambda: torch.nn.functional.linear(A, W, bb))
    fl = 2.0 * M * N * K
    best = min(t128, t256, t256_16, t128_16, tpp)
    for k, v in (("ring128", t128), ("ring256", t256), ("ring256_16", t256_16), ("ring128_16", t128_16), ("pp", tpp), ("vendor", tv), ("best", best)):
        tot[k] += v
    f = lambda t: f"{t * 1e6:8.1f} ({fl / t / 1e12:6.1f})"  # noqa: E731
    print(f"{name:30s} {f(t128):>24s} {f(t256):>24s} {f(t256_16):>24s} {f(t128_16):>24s} {f(tpp):>22s} {f(tv):>24s}   {best / tv:.3f}")
print(f"{'sum':30s} {tot['ring128'] * 1e6:24.1f} {tot['ring256'] * 1e6:24.1f} {tot['ring256_16'] * 1e6:24.1f} {tot['ring128_16'] * 1e6:24.1f} {tot['pp'] * 1e6:22.1f} {tot['vendor'] * 1e6:24.1f}   {tot['best'] / tot['vendor']:.3f}")
