# does importing locator_amd._lib BEFORE torch.distributed (what bench.py --lib does) explain the 1 %?
cd $GRAFT_REPO_ROOT
cat > /tmp/early.py <<'PY'
import sys, runpy
from locator_amd import _lib
sys.argv = ["bench.py"] + sys.argv[1:]
runpy.run_path("bench.py", run_name="__main__")
PY
for rep in 1 2 3; do for mode in plain early; do
if [ $mode = plain ]; then CMD="python3 bench.py"; elif [ $mode = early ]; then CMD="env PYTHONPATH=$GRAFT_REPO_ROOT python3 /tmp/early.py"; else CMD="python3 bench.py --lib locator_amd/liblocator_hip.so"; fi
$CMD --steps ${STEPS:-30} --warmup 5 --no-cpu-baseline --no-l1-gemm 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('$mode', round(d['value']), d['ms_per_step'], 'step_us', d['us_per_minibatch_step'], r.get('us_per_launch'))"
done; done > gpurun_out/r06_import_order.txt 2>&1
cat gpurun_out/r06_import_order.txt
