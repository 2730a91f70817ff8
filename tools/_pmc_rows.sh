cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt_rows -o k --output-format csv -- python3 $R/tools/rows_gemm_bench.py --rows 1000 --iters 20 > $R/gpurun_out/kt_rows.log 2>&1
grep -E "l1_rows|l1_reduce" $R/gpurun_out/kt_rows/k_kernel_stats.csv | cut -c1-200
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $R/gpurun_out/pmc_rows -o p --output-format csv -- python3 $R/tools/rows_gemm_bench.py --rows 1000 --iters 3 > $R/gpurun_out/pmc_rows.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS -d $R/gpurun_out/pmc_rows2 -o p --output-format csv -- python3 $R/tools/rows_gemm_bench.py --rows 1000 --iters 3 > $R/gpurun_out/pmc_rows2.log 2>&1
python3 - <<'PY'
import csv, glob, os, collections
R=os.environ["GRAFT_REPO_ROOT"]
for d in ("pmc_rows","pmc_rows2"):
    f=glob.glob(R+"/gpurun_out/"+d+"/**/*counter_collection.csv", recursive=True)
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        k=r["Kernel_Name"][:48]
        if "l1_rows" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in acc.items():
        print(k)
        for c,vals in v.items():
            print("   ",c,len(vals),sum(vals)/len(vals))
PY
