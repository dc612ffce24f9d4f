import numpy as np, torch
torch.manual_seed(0); torch.set_num_threads(1)
def fma(a,b,c): return np.float32(np.float64(a)*np.float64(b)+np.float64(c))
def blocked(a, kb):
    s=None
    for i in range(0,len(a),kb):
        acc=np.float32(0)
        for x in a[i:i+kb]: acc=fma(x,x,acc)
        s=acc if s is None else np.float32(s+acc)
    return s
def blocked_carry(a, kb):   # accumulate continues (C += A*B per K block): same as sequential
    acc=np.float32(0)
    for x in a: acc=fma(x,x,acc)
    return acc
def unroll(a,u,comb):
    acc=[np.float32(0)]*u
    n=len(a)-len(a)%u
    for i in range(0,n,u):
        for l in range(u): acc[l]=fma(a[i+l],a[i+l],acc[l])
    if comb=="tree":
        v=acc
        while len(v)>1: v=[np.float32(v[2*i]+v[2*i+1]) for i in range(len(v)//2)]
        s=v[0]
    else:
        s=acc[0]
        for x in acc[1:]: s=np.float32(s+x)
    for x in a[n:]: s=fma(x,x,s)
    return s
def tail_first(a,u):   # remainder handled first then main
    r=len(a)%u; s=np.float32(0)
    for x in a[:r]: s=fma(x,x,s)
    for x in a[r:]: s=fma(x,x,s)
    return s
for D in (100, 64, 96, 128, 300):
    B,Q,L=16,4,800
    d=torch.randn(B,L,D)*0.4; q=d[:,:Q,:].clone()
    dot=torch.bmm(q,d.transpose(1,2))
    cands={"seq":lambda a:blocked_carry(a,1)}
    for kb in (2,4,8,16,24,32,48,50,64,96): cands[f"blk{kb}"]=lambda a,kb=kb:blocked(a,kb)
    for u in (2,3,4,6,8):
        for c in ("tree","seq"): cands[f"unr{u}{c}"]=lambda a,u=u,c=c:unroll(a,u,c)
    res={}
    for n,fn in cands.items():
        ok=tot=0
        for b in range(6):
            for t in range(Q):
                a=d[b,t].numpy(); tot+=1; ok+= fn(a)==dot[b,t,t].item()
        res[n]=ok
    top=sorted(res.items(),key=lambda kv:-kv[1])[:4]
    print(D, top, "of", tot)
