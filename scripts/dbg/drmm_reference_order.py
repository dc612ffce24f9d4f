"""Which summation order gives torch's cos(a, a) bits on this machine?  (DRMM.py:62-66: `sim < 1.0` on cos(a, a))"""
import numpy as np, torch, itertools
torch.manual_seed(0)
def ref_cos(q, d):  # reference common.py:160-167 semantics: bmm / ((|q|+1e-9)(|d|+1e-9))
    dot = torch.bmm(q, d.transpose(1, 2))
    qn = torch.norm(q, p=2, dim=2).view(q.shape[0], q.shape[1], 1) + 1e-9
    dn = torch.norm(d, p=2, dim=2).view(d.shape[0], 1, d.shape[1]) + 1e-9
    return dot / (qn * dn), dot, qn, dn
def f32(x): return np.float32(x)
def seq_fma(a, b):
    acc = np.float32(0)
    for x, y in zip(a, b): acc = np.float32(np.float64(x) * np.float64(y) + np.float64(acc))
    return acc
def seq_mul_add(a, b):
    acc = np.float32(0)
    for x, y in zip(a, b): acc = np.float32(acc + np.float32(x * y))
    return acc
def lanes(a, b, W, fma=True, tree=True):
    n = len(a); acc = np.zeros(W, np.float32)
    for i in range(0, n - n % W, W):
        for l in range(W):
            acc[l] = np.float32(np.float64(a[i+l]) * np.float64(b[i+l]) + np.float64(acc[l])) if fma else np.float32(acc[l] + np.float32(a[i+l]*b[i+l]))
    v = list(acc)
    if tree:
        while len(v) > 1: v = [np.float32(v[i] + v[i + len(v)//2]) for i in range(len(v)//2)]
        s = v[0]
    else:
        s = np.float32(0)
        for x in v: s = np.float32(s + x)
    for i in range(n - n % W, n):
        s = np.float32(np.float64(a[i]) * np.float64(b[i]) + np.float64(s)) if fma else np.float32(s + np.float32(a[i]*b[i]))
    return s
cands = {"seq_fma": seq_fma, "seq_muladd": seq_mul_add}
for W in (4, 8, 16, 32):
    for fma in (True, False):
        for tree in (True, False):
            cands[f"lanes{W}_{'fma' if fma else 'ma'}_{'tree' if tree else 'seq'}"] = (lambda a, b, W=W, fma=fma, tree=tree: lanes(a, b, W, fma, tree))
for D in (50, 100, 300):
    for (B, Q, L, thr) in ((16, 4, 800, 1), (16, 4, 800, 8), (1, 4, 37, 1), (64, 4, 30, 1)):
        torch.set_num_threads(thr)
        d = torch.randn(B, L, D) * 0.4
        q = d[:, :Q, :].clone()               # query term t == doc term t: cos(a, a)
        cos, dot, qn, dn = ref_cos(q, d)
        n = min(B, 8)
        res = {}
        for name, fn in cands.items():
            ok_dot = ok_n2 = tot = 0
            for b in range(n):
                for t in range(Q):
                    a = d[b, t].numpy()
                    tot += 1
                    ok_dot += fn(a, a) == dot[b, t, t].item()
                    # norm: sqrt of sum of squares in the same order
                    ok_n2 += np.float32(np.sqrt(fn(a, a))) == np.float32(qn[b, t, 0].item() - 0)  # (+1e-9 is absorbed in fp32)
            res[name] = (ok_dot, ok_n2, tot)
        best_dot = max(res.items(), key=lambda kv: kv[1][0]); best_n = max(res.items(), key=lambda kv: kv[1][1])
        lt1 = float((torch.diagonal(cos[:, :, :Q], dim1=1, dim2=2) < 1.0).float().mean())
        print(f"D={D} B={B} L={L} thr={thr}: dot best {best_dot[0]} {best_dot[1][0]}/{best_dot[1][2]}   norm best {best_n[0]} {best_n[1][1]}/{best_n[1][2]}   P(cos(a,a)<1)={lt1:.2f}")
