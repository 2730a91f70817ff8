cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_chain.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -5
for rep in 1 2 3; do for lib in "" "--lib build/liblocator_hip_nolag.so"; do
python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-l1-gemm $lib 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('$lib' or 'product(lag)', round(d['value']), d['ms_per_step'], r['frac'], r.get('us_per_launch'), d.get('final_loss'))"
done; done > gpurun_out/r06_chain_fwdlag.txt 2>&1
cat gpurun_out/r06_chain_fwdlag.txt
