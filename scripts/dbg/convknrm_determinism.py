"""Are ConvKNRM's training routes deterministic run to run, and how far apart do the autograd and the device-kernel route end after 1..5 steps?"""
import contextlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.helpers import load_case  # noqa: E402
from tests import test_gpu_parity as T  # noqa: E402
from capreolus_amd.trainer import PytorchTrainer  # noqa: E402

name, softmax = (sys.argv[1] if len(sys.argv) > 1 else "ranklist"), True
c = load_case("convknrm", name)
B = min(32, c["query"].shape[0])
rs = np.random.RandomState(5)
batches = []
for _ in range(5):
    perm = rs.permutation(c["query"].shape[0])
    batches.append({"qid": [str(i) for i in range(B)], "query": torch.as_tensor(c["query"][:B]), "query_idf": torch.as_tensor(c["query_idf"][:B]),
                    "posdoc": torch.as_tensor(c["posdoc"][:B]), "negdoc": torch.as_tensor(c["posdoc"][perm[:B]])})


def run(fused, steps):
    r = T._convknrm_reranker(c)
    m = r.model
    m.train()
    t = PytorchTrainer({"batch": B, "itersize": steps * B, "lr": 0.01, "graph": False, "fused": fused, "softmaxloss": softmax})
    t.device, t.scaler, t._train_autocast = torch.device("cuda:0"), None, contextlib.nullcontext
    t.loss = t.pair_softmax_loss if softmax else t.pair_hinge_loss
    t._train_graph, t._graph_failed, t._fused_failed = None, False, False
    t._use_fused = t._fused_allowed(r)
    t.optimizer = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=0.01)
    t._set_lr(0)
    loss = t.single_train_iteration(r, batches[:steps], cur_iter=1)
    return float(loss), {k: v.detach().cpu().clone() for k, v in m.named_parameters() if v.requires_grad}


for steps in (1, 2, 3, 5):
    la, a = run(False, steps)
    lb, b = run(False, steps)
    lc, cc = run(True, steps)
    ld, d = run(True, steps)
    same_e = max(float((a[k] - b[k]).abs().max()) for k in a)
    same_f = max(float((cc[k] - d[k]).abs().max()) for k in a)
    cw = "convs.0.0.weight"
    diff = (a[cw] - cc[cw]).abs()
    print("steps %d: eager run-to-run max diff %.3g, fused run-to-run %.3g; eager vs fused: loss %.6f / %.6f, %s mean |diff| %.3g, frac > 1e-3: %.4f, combine.0.weight max diff %.3g"
          % (steps, same_e, same_f, la, lc, cw, float(diff.mean()), float((diff > 1e-3).float().mean()), float((a["combine.0.weight"] - cc["combine.0.weight"]).abs().max())))
