#!/bin/bash
# One rocprofv3 --pmc pass with the counters given in $PMC over `python3 tools/rows_gemm_bench.py $GEMM_BENCH_ARGS`
# (GPU box, repo root); prints per-kernel means of every counter for the l1_gemm kernels.
# At most 4 counters of one hardware block per pass (more: "exceeds the capabilities of the hardware", and rocprofv3 then
# hangs in its finaliser - hence the timeout).
#   PMC="TCC_HIT TCC_MISS" GEMM_BENCH_ARGS="--rows 4096 --i8-only --iters 3" bash tools/pmc_counters.sh
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
rm -rf $O/pmcx
timeout 200 rocprofv3 --kernel-trace --pmc $PMC -d $O/pmcx -o p --output-format csv -- python3 $R/tools/rows_gemm_bench.py ${GEMM_BENCH_ARGS:---rows 4096 --i8-only --iters 3} > $O/pmcx.log 2>&1
python3 - <<PY
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/pmcx/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "l1_gemm" in k and "reduce" not in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k, json.dumps({c: round(sum(x) / len(x), 1) for c, x in v.items()}))
PY
grep -iE "error|invalid|not found" $O/pmcx.log | head -3
