cd /root/repo
fail=0
for i in $(seq 1 12); do
  python tests/thread_capture_stress.py 2>&1 | tail -1
done
for i in 1 2 3; do
  python -m pytest tests/test_gpu_cli.py -x -q -k "bootstrap_outputs or windows_on_zarr" 2>&1 | tail -1
done
python tools/readme_windows.py --layouts default --repeat 6 2>/dev/null | python -c "
import sys,json; d=json.load(sys.stdin); print(d['layouts']['default']['walls_s'], d['layouts']['default']['rc'])"
