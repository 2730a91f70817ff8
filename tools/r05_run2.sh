cd /root/repo
python -m pytest tests/test_gpu_parity.py -x -q -k 'validation_sweep_arithmetic or refuses_a_permutation' 2>&1 | tail -40
python -m pytest tests/test_gpu_gemm_i8.py tests/test_gpu_trained_predict.py -x -q 2>&1 | tail -5
python tools/rows_gemm_bench.py --i8-only --rows 1000,4096 --iters 50 2>&1 | tail -8
for L in build/liblocator_hip_ablate64.so build/liblocator_hip_ablate128.so; do echo LIB=$L; python tools/rows_gemm_bench.py --i8-only --rows 1000 --iters 50 --lib $L 2>&1 | tail -2; done
