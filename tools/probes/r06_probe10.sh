cd $GRAFT_REPO_ROOT
for rep in 1 2 3 4; do for lib in "" "--lib build/liblocator_hip_r05chain.so"; do
python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-l1-gemm $lib 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('$lib' or 'product', round(d['value']), d['ms_per_step'], 'step_us', d['us_per_minibatch_step'], r.get('us_per_launch'), d.get('final_loss'))"
done; done
