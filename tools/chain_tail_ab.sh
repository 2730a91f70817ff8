# Chained step with its Adam tail merged into the layer-1 launch (default) against a separate tail launch, on the GPU box
for f in "" "--separate-tail"; do
  timeout 200 python bench.py --no-l1-gemm --no-cpu-baseline --steps 60 $f 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
r = d['roofline']
print('flags [$f]', 'step us', d['us_per_minibatch_step'], 'kernel us', r['us_per_launch'], 'samples/s', d['value'], 'loss', d['final_loss'])
"
done
