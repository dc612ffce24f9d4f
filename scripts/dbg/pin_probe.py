"""DataLoader pin_memory cost per batch size on the GPU box (builder-side aid)."""
import time, numpy as np, torch
N = 20000
q = np.zeros((N, 4), np.int64); d = np.zeros((N, 800), np.int64); idf = np.zeros((N, 4), np.float32)
class DS(torch.utils.data.IterableDataset):
    def __iter__(self):
        for r in range(N):
            yield {"qid": str(r // 1000), "posdocid": f"d{r}", "query": q[r], "posdoc": d[r], "query_idf": idf[r]}
torch.zeros(1, device="cuda")
for pin in (False, True):
    for bs in (32, 256, 1000):
        for rep in range(2):
            t = time.perf_counter()
            for b in torch.utils.data.DataLoader(DS(), batch_size=bs, pin_memory=pin):
                x = {k: v.to("cuda", non_blocking=True) for k, v in b.items() if torch.is_tensor(v)}
            torch.cuda.synchronize()
            dt = time.perf_counter() - t
        print("pin", pin, "bs", bs, round(dt, 3))
