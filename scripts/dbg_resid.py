import ctypes, sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from capreolus_amd import _lib
lib=_lib.load(); DEV="cuda:0"
_p=lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
st=ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def to_cm(x):
    M,C=x.shape; return x.reshape(M//32,32,C//8,8).permute(0,2,1,3).contiguous().reshape(-1)
def from_cm(f,M,C): return f.reshape(M//32,C//8,32,8).permute(0,2,1,3).contiguous().reshape(M,C)
M,N,K=256,256,128
tdt=torch.float16
A=torch.zeros((M,K),device=DEV,dtype=tdt); W=torch.zeros((N,K),device=DEV,dtype=tdt)
bp=torch.zeros(N,device=DEV)
R=(torch.arange(M,device=DEV)[:,None]*0+torch.arange(N,device=DEV)[None,:]).to(tdt)   # R[m][n] = n
gamma=torch.ones(N,device=DEV)
mr=torch.stack([torch.zeros(M,device=DEV),torch.ones(M,device=DEV)],1).contiguous()
out=torch.empty(M*N,dtype=tdt,device=DEV); part=torch.zeros((M,N//64,2),device=DEV)
rc=lib.capamd_bert_gemm_ln(_p(A),_p(W),_p(bp),M,N,K,5|0x100,None,None,None,None,_p(to_cm(R)),_p(mr),_p(gamma),_p(part),_p(out),1,st)
got=from_cm(out,M,N).float()
print("rc",rc); print("row0", got[0,:48].tolist()); print("row33", got[33,64:96].tolist())
# bias test: bias[n]=n, R=0
bp=torch.arange(N,device=DEV).float(); R0=torch.zeros((M,N),device=DEV,dtype=tdt)
rc=lib.capamd_bert_gemm_ln(_p(A),_p(W),_p(bp),M,N,K,5|0x100,None,None,None,None,_p(to_cm(R0)),_p(mr),_p(gamma),_p(part),_p(out),1,st)
got=from_cm(out,M,N).float(); print("bias row0", got[0,:48].tolist())
# acc test: A = e_0 (first k), W[n][0]=n -> acc[m][n] = n
A=torch.zeros((M,K),device=DEV,dtype=tdt); A[:,0]=1; W=torch.zeros((N,K),device=DEV,dtype=tdt); W[:,0]=torch.arange(N,device=DEV).to(tdt)
bp=torch.zeros(N,device=DEV)
rc=lib.capamd_bert_gemm_ln(_p(A),_p(W),_p(bp),M,N,K,5|0x100,None,None,None,None,_p(to_cm(R0)),_p(mr),_p(gamma),_p(part),_p(out),1,st)
got=from_cm(out,M,N).float(); print("acc row0", got[0,:48].tolist())
