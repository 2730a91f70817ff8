#!/bin/bash
# usage (GPU box, repo root): bash tools/gemm_variants.sh "<args for gemm_variants.py>"  -> per-kernel average durations
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
rm -rf $O/gv && mkdir -p $O/gv
rocprofv3 --kernel-trace --stats -d $O/gv -o k --output-format csv -- python3 $R/tools/gemm_variants.py $1 > $O/gv.log 2>&1
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$O/gv/k_kernel_stats.csv")))
for r in rows:
    n = r["Name"]
    if "l1_gemm" in n or "l1_image" in n:
        print("%-60s calls %4s avg %9.2f us  min %9.2f  max %9.2f" % (n.split("(")[0].replace("void ", ""), r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
