import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from capreolus_amd import _lib
lib=_lib.load(); dev="cuda:0"
vp=lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
st=ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
M,N,K=65536,768,768
A=torch.randn((M,K),device=dev).half(); W=(torch.randn((N,K),device=dev)*0.05).half(); bias=torch.randn(N,device=dev); out=torch.empty((M,N),dtype=torch.float16,device=dev)
for epi in (0x101, 0x100):
    stamps=torch.zeros((256,32),dtype=torch.int64,device=dev)
    for _ in range(2): lib.capamd_bert_gemm(vp(A),vp(W),vp(bias),M,N,K,epi,None,vp(out),1,st)
    lib.capamd_debug_set_gemm_stamps(vp(stamps)); lib.capamd_bert_gemm(vp(A),vp(W),vp(bias),M,N,K,epi,None,vp(out),1,st); torch.cuda.synchronize(); lib.capamd_debug_set_gemm_stamps(None)
    s=stamps.cpu().numpy()
    # last tile's epilogue internal stamps in slots 24..30; tile stamps in 0..: find last epilogue start/end
    n=int((s[0,:24]!=0).sum()); 
    e0=s[:,n-2]; e1=s[:,n-1]
    d=np.median(np.stack([s[:,24]-e0, s[:,25]-s[:,24], s[:,26]-s[:,25], s[:,28]-s[:,26], s[:,29]-s[:,28], e1-s[:,29]],1),axis=0)
    print(hex(epi), "epilogue total", int(np.median(e1-e0)), "| start->entry, vectors i=0 landed, ->j=2 (i=0), ->vectors i=1 landed, ->j=2 (i=1), ->end:", [int(x) for x in d])
