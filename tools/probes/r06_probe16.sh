# is the one-off ~1 ms inside bench.py's timed region a generation-2 garbage collection?
cd $GRAFT_REPO_ROOT
for rep in 1 2 3 4; do for mode in stats freeze disable; do
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-l1-gemm --gc-probe $mode 2>/tmp/err.txt | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('$mode', round(d['value']), d['ms_per_step'], 'step_us', d['us_per_minibatch_step'])"
grep "gc collections" /tmp/err.txt
done; done > gpurun_out/r06_gc_probe.txt 2>&1
cat gpurun_out/r06_gc_probe.txt
