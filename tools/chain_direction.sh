#!/bin/bash
# Cache-direction probe of the chained layer-1 kernel (VERDICT r05 next #1a):  bash tools/chain_direction.sh
# A = product library (every step walks its k-tiles first to last), B = `bash tools/probes/build_chain_probe.sh chalt
# -DLOC_CHAIN_ALT=1` (odd Adam steps walk them last to first, so a step starts on what the previous step touched last), interleaved,
# per cache policy (loc_tuning.l1b_nt_mask).  Prints samples/s and the event-bracketed time of the chained kernel per run.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
ALT=${ALT:-build/liblocator_hip_chalt.so}
[ -f $ALT ] || bash tools/probes/build_chain_probe.sh chalt -DLOC_CHAIN_ALT=1
for rep in 1 2 3; do
  for m in 0 -1 9 15; do
    for lib in "" "--lib $ALT"; do
      python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-l1-gemm --nt-mask $m $lib $EXTRA 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print(json.dumps({'rep':$rep,'nt_mask':$m,'lib':'$lib' or 'product','samples_per_s':round(d['value']),'ms_per_epoch':d['ms_per_step'],'chain_frac':r['frac'],'chain_TBps':r['achieved']}))"
    done
  done
done
