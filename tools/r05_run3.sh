cd /root/repo
echo "=== tests"; python -m pytest tests/test_gpu_parity.py -x -q -k 'validation_sweep_arithmetic or refuses_a_permutation' 2>&1 | tail -15
python -m pytest tests/test_gpu_gemm_i8.py tests/test_gpu_trained_predict.py tests/test_gpu_baseline_sizes.py -x -q 2>&1 | tail -15
echo "=== gemm"; python tools/rows_gemm_bench.py --i8-only --rows 1000,4096 --iters 50 2>&1 | tail -4
python tools/rows_gemm_bench.py --i8-only --rows 1000 --iters 50 --blocks 128 2>&1 | tail -2
for L in build/liblocator_hip_ablate64.so build/liblocator_hip_ablate128.so; do echo LIB=$L; python tools/rows_gemm_bench.py --i8-only --rows 1000 --iters 50 --lib $L 2>&1 | tail -2; done
echo "=== readme windows"; python tools/readme_windows.py --busy > gpurun_out/r05_readme_windows.json 2> gpurun_out/r05_readme_windows.err; tail -12 gpurun_out/r05_readme_windows.err
