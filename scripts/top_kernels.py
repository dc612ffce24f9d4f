"""Prints the top kernels of a rocprofv3 `*_kernel_stats.csv` (share of GPU time, calls, average duration)."""
import csv
import glob
import sys

f = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".csv") else glob.glob((sys.argv[1] if len(sys.argv) > 1 else ".") + "/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[: int(sys.argv[2]) if len(sys.argv) > 2 else 14]:
    print("%6.2f%% calls=%6s avg=%9.1f us  %s" % (float(r["Percentage"]), r["Calls"], float(r["AverageNs"]) / 1e3, r["Name"][:100]))
