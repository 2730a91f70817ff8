#!/bin/bash
# BASELINE.json configs[3] at full size on ONE GPU: synthetic Ag1000G-scale zarr store (765 samples,
# 25 windows of 2 Mb x ~150k variants = 3.75M variants, 5.7 GB of int8 calls), --windows --window_size 2000000.
# Writes gpurun_out/config4.log.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out /tmp/c4out
t0=$(date +%s%N)
python3 tools/make_synth_zarr.py --out /tmp/c4 --n 765 --windows ${NWIN:-25} --per_window 150000 > gpurun_out/config4.log 2>&1
t1=$(date +%s%N)
echo "store written in $(( (t1 - t0) / 1000000 )) ms" >> gpurun_out/config4.log
python3 -m locator_amd.locator --zarr /tmp/c4.zarr --sample_data /tmp/c4_samples.txt --out /tmp/c4out/win \
        --windows --window_size 2000000 --seed 12345 >> gpurun_out/config4.log 2>&1
t2=$(date +%s%N)
echo "locator --windows: wall $(( (t2 - t1) / 1000000 )) ms" >> gpurun_out/config4.log
ls /tmp/c4out | wc -l >> gpurun_out/config4.log
ls /tmp/c4out | head -4 >> gpurun_out/config4.log
tail -6 gpurun_out/config4.log
