cd /root/repo
run() { python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-l1-gemm "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['us_per_launch'])"; }
echo "default: $(run)"; echo "default: $(run)"
for G in 448 480 512 576 640 768; do echo "l1-bwd-grid $G: $(run --l1-bwd-grid $G)"; done
for M in 9 15 -1; do echo "nt-mask $M: $(run --nt-mask $M)"; done
echo "separate-tail: $(run --separate-tail)"
echo "no-xchain: $(run --no-xchain)"
echo "helpers 6: $(run --stack-helpers 6)"; echo "helpers 16: $(run --stack-helpers 16)"
