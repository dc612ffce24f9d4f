#!/bin/bash
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
for L in 1 3; do
LAYERS=$L python scripts/dbg/r5_detm2.py /tmp/new_prod.npy
LAYERS=$L PROF=1 python scripts/dbg/r5_detm2.py /tmp/new_prof.npy
LAYERS=$L CAPAMD_LIB_PATH=$R/capreolus_amd/csrc/ablate/libcapreolus_amd_nopf.so python scripts/dbg/r5_detm2.py /tmp/head_prod.npy
for pick in "qkv=128" "oproj=256x32,ffn2=256x32" "qkv=128,oproj=256x32,ffn2=256x32"; do
LAYERS=$L CAPAMD_GEMM_PICK=$pick python scripts/dbg/r5_detm2.py /tmp/p.npy; LAYERS=$L CAPAMD_GEMM_PICK=$pick PROF=1 python scripts/dbg/r5_detm2.py /tmp/q.npy
python -c "
import numpy as np
a,b=np.load('/tmp/p.npy'),np.load('/tmp/q.npy'); print('layers $L pick $pick: prod vs prof differing', int((a!=b).sum()), float(np.abs(a-b).max()))"
done
python -c "
import numpy as np
a,b,c=np.load('/tmp/new_prod.npy'),np.load('/tmp/new_prof.npy'),np.load('/tmp/head_prod.npy')
print('layers $L: new_prod vs head', int((a!=c).sum()), float(np.abs(a-c).max()), '| new_prof vs head', int((b!=c).sum()), float(np.abs(b-c).max()), '| scale', float(np.abs(c).mean()))"
done
