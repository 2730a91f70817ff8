cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for lib in "" "--lib build/liblocator_hip_il12.so"; do
python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-l1-gemm $lib 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('$lib' or 'product', round(d['value']), d['ms_per_step'], 'step_us', d['us_per_minibatch_step'], r.get('us_per_launch'), d.get('final_loss'))"
done; done > gpurun_out/r06_chain_interleaved.txt 2>&1
cat gpurun_out/r06_chain_interleaved.txt
