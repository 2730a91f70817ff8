#!/bin/bash
# PMC passes on the image-based large-M GEMMs, bf16 (l1_gemm.hip) and int8 (l1_gemm_i8.hip), on the GPU box from the
# repo root:      bash tools/gemm_pmc.sh [rows]      ->  gpurun_out/gemm_pmc_<rows>.json
# Separate rocprofv3 runs per counter set, --kernel-trace only (no --stats / --sys-trace beside --pmc).  The
# GRBM_GUI_ACTIVE pass gives the effective shader clock of each kernel: busy cycles / kernel duration from the same
# pass's kernel trace.
ROWS=${1:-1000}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
rm -rf $O/gp1 $O/gp2 $O/gp3 $O/gp4
B="python3 $R/tools/rows_gemm_bench.py --rows $ROWS --iters 3 ${GEMM_BENCH_ARGS:---gemm-only}"
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES -d $O/gp1 -o p --output-format csv -- $B > $O/gp1.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU -d $O/gp2 -o p --output-format csv -- $B > $O/gp2.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_IFETCH SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_SMEM -d $O/gp3 -o p --output-format csv -- $B > $O/gp3.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $O/gp4 -o p --output-format csv -- $B > $O/gp4.log 2>&1
python3 - <<PY
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
def short(n):
    return n.split("(")[0].replace("void ", "")
for d in ("gp1", "gp2", "gp3", "gp4"):
    for f in glob.glob("$O/" + d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if "l1_gemm" in k and "reduce" not in k:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob("$O/gp4/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if "l1_gemm" in k and "reduce" not in k:
            dur[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
out = {"rows": $ROWS, "source": "tools/gemm_pmc.sh $ROWS"}
for k, v in acc.items():
    m = {c: sum(x) / len(x) for c, x in v.items()}
    waves = 2048.0
    if "SQ_WAVE_CYCLES" in m:
        # one workgroup per CU and launch: 256 workgroups x 8 waves
        cyc = 4 * m["SQ_WAVE_CYCLES"] / waves
        f = lambda c: round(4 * m.get(c, 0) / (waves * cyc), 3)
        m["derived"] = {"cycles_per_wave": round(cyc), "mfma_busy": round(m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * cyc), 3),
                        "wait_any(parked)": f("SQ_WAIT_ANY"), "wait_inst_any(issue stall)": f("SQ_WAIT_INST_ANY"), "active_inst_any": f("SQ_ACTIVE_INST_ANY"),
                        "valu": f("SQ_ACTIVE_INST_VALU"), "lds_inst": f("SQ_ACTIVE_INST_LDS"), "vmem_inst": f("SQ_ACTIVE_INST_VMEM"), "sca": f("SQ_ACTIVE_INST_SCA"),
                        "misc": f("SQ_ACTIVE_INST_MISC"), "wait_inst_lds": f("SQ_WAIT_INST_LDS"),
                        "lds_array_busy": round(m.get("SQ_LDS_IDX_ACTIVE", 0) / (256 * cyc), 3), "lds_conflict": round(m.get("SQ_LDS_BANK_CONFLICT", 0) / max(m.get("SQ_LDS_IDX_ACTIVE", 1), 1), 3),
                        "insts_per_wave": {c: round(m.get(c, 0) / waves, 1) for c in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_IFETCH")},
                        "level_vmem(avg outstanding x cycles)": m.get("SQ_INST_LEVEL_VMEM"), "level_lds": m.get("SQ_INST_LEVEL_LDS")}
    if k in dur and "GRBM_GUI_ACTIVE" in m:
        ns = sum(dur[k]) / len(dur[k])
        m.setdefault("derived", {})["kernel_us(profiled pass)"] = round(ns * 1e-3, 2)
        # GRBM_GUI_ACTIVE is summed over the XCDs' GRBMs by rocprofv3: report both readings
        m["derived"]["clock_ghz_if_counter_is_per_device"] = round(m["GRBM_GUI_ACTIVE"] / ns, 3)
        m["derived"]["clock_ghz_if_counter_is_summed_over_8_xcds"] = round(m["GRBM_GUI_ACTIVE"] / 8 / ns, 3)
        if "cycles_per_wave" in m["derived"]:
            m["derived"]["clock_ghz_from_wave_cycles"] = round(m["derived"]["cycles_per_wave"] / ns, 3)
    out[k] = m
json.dump(out, open("$O/gemm_pmc_$ROWS${GEMM_PMC_TAG}.json", "w"), indent=1)
for k, m in out.items():
    if isinstance(m, dict):
        print(k, json.dumps(m.get("derived")))
PY
grep -iE "error|invalid|not found" $O/gp1.log $O/gp2.log $O/gp3.log $O/gp4.log | head -5
