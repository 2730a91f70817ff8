set -x
cd /root/repo
./tools/probes/splitk_reduce_probe > gpurun_out/r05_splitk_probe.jsonl 2>&1
cat gpurun_out/r05_splitk_probe.jsonl
for L in "" build/liblocator_hip_ablate64.so build/liblocator_hip_ablate128.so build/liblocator_hip_ablate1.so build/liblocator_hip_ablate2.so; do
  echo "LIB=$L"
  if [ -z "$L" ]; then python tools/rows_gemm_bench.py --i8-only --rows 1000 --iters 50 --blocks 0,128 2>&1 | grep '"digits": 2';
  else python tools/rows_gemm_bench.py --i8-only --rows 1000 --iters 50 --lib $L 2>&1 | grep '"digits": 2'; fi
done > gpurun_out/r05_gemm1000_ablate.log 2>&1
cat gpurun_out/r05_gemm1000_ablate.log
python -m pytest tests/test_gpu_parity.py -x -q -k "validation_sweep_arithmetic or refuses_a_permutation" 2>&1 | tail -5
