#!/bin/bash
# Idle fraction of a callback-driven fit: wall vs rocprofv3 kernel time (repo root on the GPU box).
#   bash tools/config1_timeline.sh <tag> [--sync]     -> gpurun_out/<tag>_config1_timeline.json, <tag>_window150k_timeline.json
TAG=${1:-r04}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for W in config1 window150k; do
  if [ $W = config1 ]; then ARGS="--fixture"; else ARGS="--n 765 --snps 150016"; fi
  python3 $R/tools/fit_timeline.py $ARGS --tag "$TAG plain $*" "$@" > $O/${TAG}_${W}_plain.json 2> $O/${TAG}_${W}.err
  rm -rf $O/kt_$W
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/kt_$W -o k --output-format csv -- python3 $R/tools/fit_timeline.py $ARGS --tag "$TAG profiled $*" "$@" > $O/${TAG}_${W}_prof.json 2>> $O/${TAG}_${W}.err
  python3 $R/tools/fit_timeline.py --report $O/${TAG}_${W}_prof.json --stats-csv $O/kt_$W/k_kernel_stats.csv > $O/${TAG}_${W}_timeline.json
  cat $O/${TAG}_${W}_plain.json $O/${TAG}_${W}_timeline.json
done
