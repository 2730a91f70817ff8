# L2 warm-up helper workgroups of the fused hidden stack under the chained schedule (bench.py --stack-helpers), on the GPU box
for h in 12 8 16 20 24; do
  timeout 200 python bench.py --no-l1-gemm --no-cpu-baseline --steps 60 --stack-helpers $h 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('helpers', $h, 'step us', d['us_per_minibatch_step'], 'samples/s', d['value'])
"
done
