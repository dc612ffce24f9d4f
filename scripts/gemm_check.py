import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from capreolus_amd import _lib
lib = _lib.load(); dev = "cuda:0"
vp = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for (M, N, K) in [(64,64,64),(64,64,128),(64,64,192),(64,64,256),(128,192,320),(64,128,128),(256,256,64),(256,256,128),(256,256,192),(256,256,320),(512,512,256),(1024,768,128),(2048,2304,768),(4096,3072,768),(8192,768,3072),(66048,768,704)]:
    for epi in (0, 1, 4):
        torch.manual_seed(M + N + K)
        A = (torch.randn((M, K), device=dev) * 0.5).bfloat16()
        W = (torch.randn((N, K), device=dev) * 0.05).bfloat16()
        bias = torch.randn(N, device=dev); resid = torch.randn((M, N), device=dev).bfloat16()
        out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        errs = []
        for rep in range(3):
            out.zero_()
            assert lib.capamd_bert_gemm(vp(A), vp(W), vp(bias), M, N, K, epi, vp(resid), vp(out), 0, st) == 0
            ref = A.float() @ W.float().t() + bias + (resid.float() if epi == 4 else 0)
            if epi == 1: ref = torch.nn.functional.gelu(ref)
            errs.append(float((out.float() - ref).abs().max()))
        print(M, N, K, "epi", epi, "max abs err", ["%.3g" % e for e in errs])
