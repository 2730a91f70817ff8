#!/bin/bash
# EXTRA="--host_filter" re-runs with the rounds 1-3 host filter (the predlocs digest must not change).
# BASELINE.json configs[3] at full size on ONE GPU with 1, 2, 4, 8, 16 worker processes (tools/run_config4.sh's store):
# wall per worker count, the scheduler's own timeline (parent prologue, worker start-up, unit work, serial part) and its
# Amdahl projection for 1..8 GPUs.  Writes gpurun_out/config4_workers.txt.   bash tools/run_config4_workers.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
O=${OUT:-gpurun_out/config4_workers.txt}
if [ ! -d /tmp/c4.zarr ]; then
t0=$(date +%s%N)
python3 tools/make_synth_zarr.py --out /tmp/c4 --n 765 --windows ${NWIN:-25} --per_window 150000 > $O 2>&1
t1=$(date +%s%N)
echo "store written in $(( (t1 - t0) / 1000000 )) ms" >> $O
fi
echo "##### EXTRA='$EXTRA'" >> $O
for W in ${WORKERS:-2 1 4 8 16}; do
    rm -rf /tmp/c4out; mkdir -p /tmp/c4out
    t1=$(date +%s%N)
    python3 -m locator_amd.locator --zarr /tmp/c4.zarr --sample_data /tmp/c4_samples.txt --out /tmp/c4out/win \
            --windows --window_size 2000000 --seed 12345 --gpus 1 --fits_per_gpu $W --keras_verbose 0 $EXTRA > /tmp/c4_w$W.log 2>&1
    rc=$?
    t2=$(date +%s%N)
    echo "=== fits_per_gpu $W: rc $rc, wall $(( (t2 - t1) / 1000000 )) ms, $(ls /tmp/c4out | grep -c predlocs) predlocs files" >> $O
    grep -E "replicate phases|replicate timeline|units:|Amdahl|workers:|FAILED" /tmp/c4_w$W.log >> $O
    md5sum /tmp/c4out/*predlocs.txt | awk '{print $1}' | md5sum | awk '{print "    predlocs digest " $1}' >> $O
done
cat $O
