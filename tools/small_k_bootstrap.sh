# --bootstrap on the reference's example data (450 samples x 5,830 SNPs after filters), 60 replicates + the FULL fit, default epochs:
# wall of the command for --fits_per_gpu 0 (auto: 3 at this size), 2 (rounds 1-4) and 4
cd /root/repo
for F in 0 2 4; do
  rm -rf /tmp/bs_$F; mkdir -p /tmp/bs_$F
  t0=$(date +%s%N)
  python -m locator_amd.locator --vcf tests/golden/test_genotypes.vcf.gz --sample_data tests/golden/test_sample_data.txt --out /tmp/bs_$F/b --bootstrap --nboots 60 --seed 12345 --keras_verbose 0 --plot_history "" --fits_per_gpu $F > /tmp/bs_$F.log 2>&1
  t1=$(date +%s%N)
  echo "fits_per_gpu $F: wall $(( (t1 - t0) / 1000000 )) ms, $(ls /tmp/bs_$F | grep -c predlocs) predlocs, digest $(cat /tmp/bs_$F/*predlocs.txt | md5sum | cut -c1-12)"
  grep -E "replicate phases|replicate timeline" /tmp/bs_$F.log | cut -c1-230
done
