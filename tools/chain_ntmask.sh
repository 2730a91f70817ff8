# Cache-policy masks of the chained layer-1 kernel (loc_tuning.l1b_nt_mask via bench.py --nt-mask), on the GPU box
for m in 13 9 15 -1; do
  timeout 200 python bench.py --no-l1-gemm --steps 60 --nt-mask $m 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
r = d['roofline']
print('ntmask', '$m', 'step us', d['us_per_minibatch_step'], 'kernel us', r['us_per_launch'], 'samples/s', d['value'])
"
done
