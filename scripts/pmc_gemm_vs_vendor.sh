#!/bin/bash
# PMC pass over scripts/gemm_bench.py: per-dispatch clock (GRBM_GUI_ACTIVE / 8 XCDs / duration) and MFMA-busy share for
# our GEMM and the hipBLASLt kernel on the same shapes.  Output: gpurun_out/gemm_vs_vendor.txt
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp; rm -rf /tmp/pg
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/pg -o g -- python $R/scripts/gemm_bench.py > /dev/null 2>&1
f=$(find /tmp/pg -name "*counter_collection.csv")
python - "$f" <<'PY' | tee $R/gpurun_out/gemm_vs_vendor.txt
import csv, collections, sys
d = collections.defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1])):
    v = d[r['Dispatch_Id']]; v[r['Counter_Name']] = float(r['Counter_Value']); v['dur'] = int(r['End_Timestamp']) - int(r['Start_Timestamp']); v['k'] = r['Kernel_Name']; v['wg'] = r['Workgroup_Size']; v['grid'] = r['Grid_Size']
agg = collections.defaultdict(list)
for v in d.values():
    if v['dur'] > 50000 and 'GRBM_GUI_ACTIVE' in v: agg[(v['k'][:70], v['grid'])].append(v)
for k, vs in agg.items():
    n = len(vs); g = sum(x['GRBM_GUI_ACTIVE'] for x in vs) / 8; dur = sum(x['dur'] for x in vs)
    print(k, 'n=%d dur=%.0fus clk=%.2fGHz mfma_busy=%.3f lds_idx/cu-cycle=%.3f bank_conflict/lds_idx=%.3f' % (
        n, dur / n / 1e3, g / dur, sum(x['SQ_VALU_MFMA_BUSY_CYCLES'] for x in vs) / (g * 256 * 4),
        sum(x.get('SQ_LDS_IDX_ACTIVE', 0) for x in vs) / (g * 256), sum(x.get('SQ_LDS_BANK_CONFLICT', 0) for x in vs) / max(1, sum(x.get('SQ_LDS_IDX_ACTIVE', 0) for x in vs))))
PY
