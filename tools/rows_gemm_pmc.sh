#!/bin/bash
# PMC study of the large-M layer-1 forward (run on the GPU box from the repo root):
#   bash tools/rows_gemm_pmc.sh   ->  gpurun_out/rows_gemm_pmc.json  (+ kernel stats csv)
# Counter passes are separate rocprofv3 runs with --kernel-trace only.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/rows_kt -o k --output-format csv -- python3 $R/tools/rows_gemm_bench.py --rows 1000,90 --iters 20 > $O/rows_kt.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $O/rows_pmc1 -o p --output-format csv -- python3 $R/tools/rows_gemm_bench.py --rows 1000 --iters 3 > $O/rows_pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 -d $O/rows_pmc2 -o p --output-format csv -- python3 $R/tools/rows_gemm_bench.py --rows 1000 --iters 3 > $O/rows_pmc2.log 2>&1
python3 - <<'PY'
import csv, glob, json, os, collections
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
O = R + "/gpurun_out"
out = {"source": "tools/rows_gemm_pmc.sh: rocprofv3 --kernel-trace --pmc (two passes) and --kernel-trace --stats on "
                 "tools/rows_gemm_bench.py, 1000 rows x 100000 SNPs x 256 units",
       "note": "means per launch, summed over all SQs. Per wave: cycles = 4*SQ_WAVE_CYCLES/waves; MFMA-busy frac = "
               "SQ_VALU_MFMA_BUSY_CYCLES/(1024 SIMDs * cycles); VALU frac = 4*SQ_ACTIVE_INST_VALU/(waves*cycles); "
               "wait frac = 4*SQ_WAIT_INST_ANY/(waves*cycles).", "kernels": {}}
for d in ("rows_pmc1", "rows_pmc2"):
    for f in glob.glob(O + "/" + d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if "l1_rows" in k or "l1_reduce" in k or "l1_gemm" in k or "l1_image" in k:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            out["kernels"].setdefault(k, {}).update({c: sum(x) / len(x) for c, x in v.items()})
for k, v in out["kernels"].items():
    if ("l1_rows" in k or "l1_gemm_kernel" in k) and "SQ_WAVE_CYCLES" in v:
        waves = 2048.0
        cyc = 4 * v["SQ_WAVE_CYCLES"] / waves
        v["derived"] = {"cycles_per_wave": cyc,
                        "mfma_busy_frac": v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * cyc),
                        "valu_frac": 4 * v.get("SQ_ACTIVE_INST_VALU", 0) / (waves * cyc),
                        "wait_inst_frac": 4 * v.get("SQ_WAIT_INST_ANY", 0) / (waves * cyc),
                        "lds_busy_frac": v.get("SQ_LDS_IDX_ACTIVE", 0) / (256 * cyc),
                        "lds_conflict_share": v.get("SQ_LDS_BANK_CONFLICT", 0) / max(v.get("SQ_LDS_IDX_ACTIVE", 1), 1)}
json.dump(out, open(O + "/rows_gemm_pmc.json", "w"), indent=1)
for k, v in out["kernels"].items():
    if "derived" in v:
        print(k, json.dumps(v["derived"]))
PY
grep -E "l1_rows|l1_reduce|l1_gemm|l1_image" $O/rows_kt/k_kernel_stats.csv | cut -c1-160
