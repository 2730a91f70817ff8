# code-placement probe: the chained kernel shifted by 4 / 8 / 12 / 20 bytes of s_nop at its entry
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for lib in "" pad1 pad2 pad3 pad5; do
L=""; [ -n "$lib" ] && L="--lib build/liblocator_hip_$lib.so"
python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-l1-gemm $L 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('$lib' or 'product', round(d['value']), d['ms_per_step'], 'step_us', d['us_per_minibatch_step'], r.get('us_per_launch'))"
done; done > gpurun_out/r06_chain_placement.txt 2>&1
cat gpurun_out/r06_chain_placement.txt
