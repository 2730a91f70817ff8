#!/bin/bash
# BASELINE.json configs[4] at full size on ONE GPU: 1000 individuals x ~500k SNPs, --bootstrap --nboots 256
# (257 fits, two per GPU).  Writes gpurun_out/config5.log.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out /tmp/c5out
t0=$(date +%s%N)
python3 tools/make_synth_zarr.py --out /tmp/c5 --n 1000 --windows 1 --per_window 560000 > gpurun_out/config5.log 2>&1
t1=$(date +%s%N)
echo "store written in $(( (t1 - t0) / 1000000 )) ms" >> gpurun_out/config5.log
python3 -m locator_amd.locator --zarr /tmp/c5.zarr --sample_data /tmp/c5_samples.txt --out /tmp/c5out/boot \
        --bootstrap --nboots ${NBOOTS:-256} --seed 12345 >> gpurun_out/config5.log 2>&1
t2=$(date +%s%N)
echo "locator --bootstrap --nboots ${NBOOTS:-256}: wall $(( (t2 - t1) / 1000000 )) ms" >> gpurun_out/config5.log
ls /tmp/c5out | wc -l >> gpurun_out/config5.log
head -3 /tmp/c5out/boot_bootFULL_predlocs.txt >> gpurun_out/config5.log 2>&1
echo "predlocs files: $(ls /tmp/c5out | grep -c predlocs)" >> gpurun_out/config5.log
tail -5 gpurun_out/config5.log
