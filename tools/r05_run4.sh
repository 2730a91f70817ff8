cd /root/repo
R=/root/repo; O=$R/gpurun_out
echo "=== tests"; python -m pytest tests/test_gpu_gemm_i8.py tests/test_gpu_trained_predict.py tests/test_gpu_baseline_sizes.py tests/test_gpu_gemm.py -x -q 2>&1 | tail -8
cd /tmp && export TMPDIR=/tmp
rm -rf $O/gemm_kt
rocprofv3 --kernel-trace --stats -d $O/gemm_kt -o k --output-format csv -- python3 $R/tools/rows_gemm_bench.py --i8-only --rows 1000 --iters 20 > $O/r05_gemm_bench_a.log 2>&1
cut -d, -f1-6 $O/gemm_kt/k_kernel_stats.csv | head -14
tail -3 $O/r05_gemm_bench_a.log
