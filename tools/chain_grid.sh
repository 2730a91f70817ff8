# Workgroup count of the chained layer-1 kernel (bench.py --l1-bwd-grid = 2 x workgroups), on the GPU box
for g in 512 482 496 448; do
  timeout 200 python bench.py --no-l1-gemm --no-cpu-baseline --steps 60 --l1-bwd-grid $g 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
r = d['roofline']
print('chain workgroups', $g // 2, 'step us', d['us_per_minibatch_step'], 'kernel us', r['us_per_launch'], 'samples/s', d['value'])
"
done
