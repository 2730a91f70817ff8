# Infinity-Cache warm-up probe: percent of (m, v) read right before the chained kernel; us_per_launch = the chained kernel alone
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for lib in "" warm0m warm50nt warm50 warm75 warm100; do
L=""; [ -n "$lib" ] && L="--lib build/liblocator_hip_$lib.so"
python3 bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-l1-gemm $L 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('$lib' or 'product', 'chain_us', r.get('us_per_launch'), 'step_us', d['us_per_minibatch_step'], 'samples/s', round(d['value']), d.get('final_loss'))"
done; done > gpurun_out/r06_warm_probe.txt 2>&1
cat gpurun_out/r06_warm_probe.txt
