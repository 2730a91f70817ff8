cd $GRAFT_REPO_ROOT
python3 tools/probes/chain_stamps.py build/liblocator_hip_chstamps.so > gpurun_out/r06_chain_stamps.txt 2>gpurun_out/r06_chain_stamps.err
for rep in 1 2 3; do for lib in "" "--lib build/liblocator_hip_chstag2.so" "--lib build/liblocator_hip_chstag4.so"; do
python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-l1-gemm $lib 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('$lib' or 'product', round(d['value']), d['ms_per_step'], r['frac'])"
done; done > gpurun_out/r06_chain_stagger.txt 2>&1
cat gpurun_out/r06_chain_stamps.txt; tail -3 gpurun_out/r06_chain_stamps.err; cat gpurun_out/r06_chain_stagger.txt
