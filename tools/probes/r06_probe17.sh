cd $GRAFT_REPO_ROOT
for rep in 1 2 3 4 5; do
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-l1-gemm --epoch-times 2>/tmp/err.txt | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('plain', round(d['value']), d['ms_per_step'])"
grep "epoch completion" /tmp/err.txt
done > gpurun_out/r06_epoch_times.txt 2>&1
cat gpurun_out/r06_epoch_times.txt
