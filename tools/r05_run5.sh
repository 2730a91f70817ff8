cd /root/repo
R=/root/repo; O=$R/gpurun_out
echo "=== tests"; python -m pytest tests/test_gpu_stack_rows.py tests/test_gpu_gemm_i8.py tests/test_gpu_trained_predict.py tests/test_gpu_baseline_sizes.py tests/test_gpu_gemm.py -x -q 2>&1 | tail -8
echo "=== stack rows"; python tools/stack_rows_bench.py 2>&1 | tail -12 | tee $O/r05_stack_rows_bench.jsonl
cd /tmp && export TMPDIR=/tmp
rm -rf $O/gemm_kt
rocprofv3 --kernel-trace --stats -d $O/gemm_kt -o k --output-format csv -- python3 $R/tools/rows_gemm_bench.py --i8-only --rows 1000 --iters 20 > $O/r05_gemm_bench_a.log 2>&1
python3 - <<'PY'
import csv
for r in csv.DictReader(open('/root/repo/gpurun_out/gemm_kt/k_kernel_stats.csv')):
    print(r['Name'][:50].ljust(52), r['Calls'], r['AverageNs'], r['MinNs'])
PY
tail -3 $O/r05_gemm_bench_a.log
