for k in 50000 100000 200000; do echo "== snps $k"; bash tools/gemm_variants.sh "--variants 0,30,14,516 --pieces 1 --snps $k" | grep -E "l1_gemm"; grep "variant 516" gpurun_out/gv.log; done
