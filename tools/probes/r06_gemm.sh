cd $GRAFT_REPO_ROOT
python3 tools/gemm_packed_crossover.py --out gpurun_out/r06_gemm_packed_crossover.jsonl 2> gpurun_out/r06_gemm_packed_crossover.err
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -4
