cd /root/repo
export OUT=gpurun_out/r05_config4_workers.txt
rm -f $OUT
WORKERS="2" EXTRA="" bash tools/run_config4_workers.sh > /dev/null 2>&1
WORKERS="2" EXTRA="--no_graph" bash tools/run_config4_workers.sh > /dev/null 2>&1
WORKERS="2 1" EXTRA="--in_process" bash tools/run_config4_workers.sh > /dev/null 2>&1
cat $OUT
python -m pytest tests/test_gpu_parity.py -x -q -k "two_fit_threads" 2>&1 | tail -3
