# configs[3] on one GPU: worker start methods (forkserver = default, spawn = rounds 1-5, in-process) - timeline + Amdahl projection
cd $GRAFT_REPO_ROOT
export OUT=gpurun_out/r06_config4_workers.txt
rm -f $OUT
for rep in 1 2; do
WORKERS="0" EXTRA="" bash tools/run_config4_workers.sh > /dev/null 2>&1
WORKERS="0" EXTRA="--worker_start spawn" bash tools/run_config4_workers.sh > /dev/null 2>&1
WORKERS="0" EXTRA="--in_process" bash tools/run_config4_workers.sh > /dev/null 2>&1
done
cat $OUT
grep -h "cpu_affinity\|affinity" /tmp/c4_w0.log | head -3
