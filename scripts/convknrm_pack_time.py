import time, torch, sys
sys.path.insert(0, '/root/repo')
from capreolus_amd import engine
dev = torch.device('cuda:0')
V, D, F = 400001, 300, 128
emb = torch.randn((V, D), device=dev) * 0.4
ws = [torch.randn((F, D, g), device=dev) * 0.05 for g in (1, 2, 3)]
bs = [torch.randn(F, device=dev) * 0.1 for _ in range(3)]
for i in range(2):
    t = engine.ConvProjectionTables()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tab = t.get(emb, ws, bs)
    torch.cuda.synchronize(); print('pack tables: %.1f ms' % ((time.perf_counter() - t0) * 1e3), tab.numel() * 4 / 1e9, 'GB')
