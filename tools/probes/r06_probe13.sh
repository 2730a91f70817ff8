# order-bias check of the placement probe: the product library also as --lib, at three positions of the rotation
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for lib in pad5 "" pad1 same pad3 same; do
L=""; [ -n "$lib" ] && L="--lib build/liblocator_hip_$lib.so"
python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-l1-gemm $L 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('$lib' or 'product', round(d['value']), d['ms_per_step'], 'step_us', d['us_per_minibatch_step'], r.get('us_per_launch'))"
done; done > gpurun_out/r06_chain_placement2.txt 2>&1
cat gpurun_out/r06_chain_placement2.txt
