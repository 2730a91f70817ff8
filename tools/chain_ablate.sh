# Timing ablations of the chained layer-1 kernel on the GPU box: build the variants first, e.g.
#   for a in 1 2 4 16 32; do make -C locator_amd/csrc ablate_chain A=$a; done
# (LOC_CHAIN_ABLATE bits in locator_amd/csrc/l1_chain.hip; results are wrong by construction, only the time counts)
for a in "" 1 2 4 16 32; do
  if [ -z "$a" ]; then L=""; else L="--lib build/liblocator_hip_chain$a.so"; fi
  timeout 200 python bench.py --no-l1-gemm --steps 60 $L 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
r = d['roofline']
print('ablate', '$a' or 0, 'step us', d['us_per_minibatch_step'], 'kernel us', r['us_per_launch'], 'samples/s', d['value'], 'loss', d['final_loss'], d['final_val_loss'])
"
done
