# is "bench.py --lib X" itself faster than "bench.py" for the SAME library?  (probe13 suggested 0.7 %)
cd $GRAFT_REPO_ROOT
for rep in 1 2 3 4; do for lib in "" "--lib locator_amd/liblocator_hip.so" "--lib build/liblocator_hip_same.so" ""; do
python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-l1-gemm $lib 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('$lib' or 'no-flag', round(d['value']), d['ms_per_step'], 'step_us', d['us_per_minibatch_step'], r.get('us_per_launch'))"
done; done > gpurun_out/r06_lib_flag.txt 2>&1
cat gpurun_out/r06_lib_flag.txt
