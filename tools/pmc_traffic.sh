#!/bin/bash
# HBM traffic of the training kernels from PMC counters (run on the GPU box from the repo root):
#   bash tools/pmc_traffic.sh  ->  gpurun_out/pmc_traffic.json
# FETCH_SIZE and WRITE_SIZE are collected in separate rocprofv3 passes with --kernel-trace only and
# corrected / calibrated as tools/pmc_traffic.py describes (MI355X_MICROARCH.md, HBM section).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_f -o f --output-format csv -- python3 $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-l1-gemm --no-graph > $O/pmc_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_w -o w --output-format csv -- python3 $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-l1-gemm --no-graph > $O/pmc_w.log 2>&1
python3 $R/tools/pmc_traffic.py $(find $O/pmc_f -name "*counter_collection.csv" | head -1) $(find $O/pmc_w -name "*counter_collection.csv" | head -1) $O/pmc_traffic.json 100000 256 1000
