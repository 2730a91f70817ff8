cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_chain.py tests/test_gpu_parity.py tests/test_gpu_edge.py tests/test_gpu_baseline_sizes.py -x -q -m gpu 2>&1 | tail -4
for rep in 1 2 3; do for fl in "" "--reduce-launch"; do
python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-l1-gemm $fl 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('$fl' or 'fused-reduce (default)', round(d['value']), d['ms_per_step'], 'step_us', d['us_per_minibatch_step'], r.get('us_per_launch'), d.get('final_loss'))"
done; done > gpurun_out/r06_fused_reduce.txt 2>&1
cat gpurun_out/r06_fused_reduce.txt
