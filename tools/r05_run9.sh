cd /root/repo
for K in 500000 150016; do
for M in 0 15 9 -1; do
  echo "K=$K nt-mask=$M: $(python bench.py --snps $K --n 1000 --steps 8 --warmup 3 --nt-mask $M --no-cpu-baseline --no-l1-gemm 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['us_per_launch'], d['roofline']['frac'], d['whole_step_hbm_frac'])")"
done; done
