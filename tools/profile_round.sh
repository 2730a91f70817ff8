#!/bin/bash
# Regenerates the round's profile evidence on the GPU box (repo root):  bash tools/profile_round.sh r02
#   gpurun_out/<tag>_bench_default.json          python3 bench.py --steps 20 --warmup 5          (the driver's command)
#   gpurun_out/<tag>_bench_kernel_stats.csv      rocprofv3 --kernel-trace --stats of the same command
#   gpurun_out/<tag>_bench_profiled.json         the bench line printed under the profiler
#   gpurun_out/<tag>_bench_replicates2.json      python3 bench.py --replicates-per-gpu 2
#   gpurun_out/<tag>_bench_2ranks_selflaunch.json  python3 bench.py --gpus 2 --device-index 0 --dist-backend gloo (no launcher)
#   gpurun_out/<tag>_pmc_traffic.json            FETCH_SIZE / WRITE_SIZE passes (tools/pmc_traffic.sh)
#   gpurun_out/<tag>_chain_pmc.json              SQ counters of the chained layer-1 kernel (tools/chain_pmc.sh)
#   gpurun_out/<tag>_gemm_pmc_1000.json, _4096.json, <tag>_gemm_kernel_stats.csv   large-M GEMM counters (with the effective
#                                                 clock from GRBM_GUI_ACTIVE) and kernel times, int8 and bf16 kernels
# Copy what should be judged into profiles/ (tracked).
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
# HBM traffic first: bench.py reports roofline.traffic from profiles/<tag>_pmc_traffic.json when its source hash matches
bash $R/tools/pmc_traffic.sh > $O/pmc_traffic.log 2>&1
cp $O/pmc_traffic.json $O/${TAG}_pmc_traffic.json
cp $O/pmc_traffic.json $R/profiles/${TAG}_pmc_traffic.json
cd $R
python3 bench.py --steps 20 --warmup 5 > $O/${TAG}_bench_default.json 2> $O/${TAG}_bench_default.err
python3 bench.py --steps 50 --warmup 5 --replicates-per-gpu 2 --no-cpu-baseline --no-l1-gemm > $O/${TAG}_bench_replicates2.json 2>> $O/${TAG}_bench_default.err
python3 bench.py --gpus 2 --device-index 0 --dist-backend gloo --steps 20 --warmup 5 > $O/${TAG}_bench_2ranks_selflaunch.json 2>> $O/${TAG}_bench_default.err
cd /tmp
rm -rf $O/prof_kt; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/prof_kt -o k --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/${TAG}_bench_profiled.json 2> $O/prof_kt.err
cp $O/prof_kt/k_kernel_stats.csv $O/${TAG}_bench_kernel_stats.csv
bash $R/tools/chain_pmc.sh > $O/chain_pmc.log 2>&1               # counters of the chained layer-1 kernel alone
cp $O/chain_pmc.json $O/${TAG}_chain_pmc.json
bash $R/tools/gemm_pmc.sh 1000 > $O/gemm_pmc.log 2>&1
bash $R/tools/gemm_pmc.sh 4096 >> $O/gemm_pmc.log 2>&1
cp $O/gemm_pmc_1000.json $O/${TAG}_gemm_pmc_1000.json
cp $O/gemm_pmc_4096.json $O/${TAG}_gemm_pmc_4096.json
rm -rf $O/gemm_kt
rocprofv3 --kernel-trace --stats -d $O/gemm_kt -o k --output-format csv -- python3 $R/tools/rows_gemm_bench.py --rows 1000,4096 --iters 20 > $O/${TAG}_gemm_bench.log 2>&1
cp $O/gemm_kt/k_kernel_stats.csv $O/${TAG}_gemm_kernel_stats.csv
python3 $R/tools/rows_gemm_bench.py --rows 4096,16384 --iters 10 --i8-only --packed >> $O/${TAG}_gemm_bench.log 2>&1    # 2-bit packed genotypes
# round 5 evidence: the full predict-mode x shape GEMM sweep (out of the bench line since round 5), the hidden-stack forms over
# row counts, the split-K hand-over probe, the README windows example in every replicate layout
python3 $R/tools/l1_gemm_sweep.py --out $O/${TAG}_l1_gemm_sweep.json > $O/l1_gemm_sweep.log 2>&1
python3 $R/tools/stack_rows_bench.py 2>/dev/null | grep rows > $O/${TAG}_stack_rows_bench.jsonl
( cd $R/tools/probes && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 splitk_reduce_probe.hip -o splitk_reduce_probe 2>/dev/null; ./splitk_reduce_probe > $O/${TAG}_splitk_probe.jsonl 2>&1 )
( cd $R && python3 tools/readme_windows.py --busy --repeat 2 > $O/${TAG}_readme_windows.json 2> $O/readme_windows.err )
# round 4 evidence: trained-weight tolerance of the predict modes, quantisation study, fit timelines, predict timeline
cd $R
python3 -m pytest tests/test_gpu_trained_predict.py -q -s > $O/${TAG}_trained_predict.log 2>&1
python3 tools/quant_study.py --out $O/${TAG}_quant_study.jsonl > $O/quant_study.log 2>&1
bash tools/config1_timeline.sh ${TAG} > $O/config1_timeline.log 2>&1
bash tools/config1_timeline.sh ${TAG}sync --sync >> $O/config1_timeline.log 2>&1
python3 tools/predict_timeline.py --mode auto > $O/${TAG}_predict_timeline.jsonl 2>/dev/null
for w in 512 128 64; do python3 bench.py --steps 20 --warmup 4 --width $w --no-cpu-baseline --no-l1-gemm > $O/${TAG}_bench_width$w.json 2>/dev/null; python3 bench.py --steps 20 --warmup 4 --width $w --no-chain --no-cpu-baseline --no-l1-gemm > $O/${TAG}_bench_width${w}_unchained.json 2>/dev/null; done
for b in 64 128; do python3 bench.py --steps 20 --warmup 4 --batch $b --no-cpu-baseline --no-l1-gemm > $O/${TAG}_bench_batch$b.json 2>/dev/null; done
python3 bench.py --steps 20 --warmup 4 --batch 64 --no-chain --no-cpu-baseline --no-l1-gemm > $O/${TAG}_bench_batch64_unchained.json 2>/dev/null
python3 bench.py --steps 40 --warmup 5 --sync-epochs --no-cpu-baseline --no-l1-gemm > $O/${TAG}_bench_sync_epochs.json 2>/dev/null
tail -c 1500 $O/${TAG}_bench_default.json; echo; cat $O/${TAG}_bench_replicates2.json | cut -c1-200; echo; cat $O/${TAG}_bench_2ranks_selflaunch.json | cut -c1-300; echo
head -12 $O/${TAG}_bench_kernel_stats.csv | cut -c1-150
tail -6 $O/pmc_traffic.log; tail -4 $O/gemm_pmc.log | cut -c1-400
# round 6 evidence (profiles/r06_*): the probes of the chained layer-1 kernel (each a one-translation-unit measurement build next
# to the product library: cache direction, cycle stamps), the packed-genotype crossover of the int8 GEMM (interleaved medians),
# configs[3] with forkserver / spawn / in-process workers.  The stagger / early-request / lagged-forward probes of round 6 were
# run from tools/probes/r06_probe*.sh on the commits named in docs/history/round6.md.
bash tools/probes/build_chain_probe.sh chalt -DLOC_CHAIN_ALT=1 > /dev/null 2>&1
bash tools/probes/build_chain_probe.sh chstamps -DLOC_CHAIN_STAMPS=100 > /dev/null 2>&1
bash tools/chain_direction.sh > $O/${TAG}_chain_direction.jsonl 2>/dev/null
python3 tools/probes/chain_stamps.py build/liblocator_hip_chstamps.so > $O/${TAG}_chain_stamps.txt 2>/dev/null
python3 tools/gemm_packed_crossover.py --out $O/${TAG}_gemm_packed_crossover.jsonl > /dev/null 2>&1
OUT=gpurun_out/${TAG}_config4_workers.txt bash tools/probes/r06_config4.sh > /dev/null 2>&1
