#!/bin/bash
# PMC passes on the many-row hidden-stack kernels (stack_rows.hip), on the GPU box:
#   bash tools/stack_rows_pmc.sh                      32-row form at 16,384 rows (512 workgroups x 8 waves) -> gpurun_out/stack_rows_pmc.json
#   ROWS=4096 TILE=16 bash tools/stack_rows_pmc.sh    16-row form (round 5) at 4096 rows (256 workgroups)
# Separate rocprofv3 runs per counter set, --kernel-trace only beside --pmc; every run under `timeout`.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
rm -rf $O/sp1 $O/sp2 $O/sp3
B="python3 $R/tools/predict_timeline.py --mode auto --rows ${ROWS:-16384}"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES -d $O/sp1 -o p --output-format csv -- $B > $O/sp1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_F32 -d $O/sp2 -o p --output-format csv -- $B > $O/sp2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE -d $O/sp3 -o p --output-format csv -- $B > $O/sp3.log 2>&1
python3 - <<PY
import csv, glob, collections, json
TILE = int("${TILE:-32}")
KPAT = "stack_rows16_eval" if TILE == 16 else "stack_rows_eval"
acc = collections.defaultdict(list)
dur = []
for d in ("sp1", "sp2", "sp3"):
    for f in glob.glob("$O/" + d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if KPAT in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob("$O/sp3/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if KPAT in r["Kernel_Name"]:
            dur.append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
m = {c: sum(x) / len(x) for c, x in acc.items()}
rows = int("${ROWS:-16384}")
waves = 8.0 * ((rows + TILE - 1) // TILE)
out = {"source": "tools/stack_rows_pmc.sh (tools/predict_timeline.py --mode auto --rows %d; 100,000 SNPs)" % rows, "counters": m}
if "SQ_WAVE_CYCLES" in m:
    cyc = 4 * m["SQ_WAVE_CYCLES"] / waves
    f = lambda c: round(4 * m.get(c, 0) / (waves * cyc), 3)
    out["derived"] = {"waves": waves, "cycles_per_wave": round(cyc),
                      "mfma_busy_cycles_per_wave_cycle": round(m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(m.get("SQ_BUSY_CYCLES", 1), 1), 3),
                      "wait_any(parked)": f("SQ_WAIT_ANY"), "wait_inst_any(issue stall)": f("SQ_WAIT_INST_ANY"),
                      "active_inst_any": f("SQ_ACTIVE_INST_ANY"), "valu": f("SQ_ACTIVE_INST_VALU"), "lds_inst": f("SQ_ACTIVE_INST_LDS"),
                      "vmem_inst": f("SQ_ACTIVE_INST_VMEM"), "misc": f("SQ_ACTIVE_INST_MISC"), "wait_inst_lds": f("SQ_WAIT_INST_LDS"),
                      "lds_conflict": round(m.get("SQ_LDS_BANK_CONFLICT", 0) / max(m.get("SQ_LDS_IDX_ACTIVE", 1), 1), 3),
                      "insts_per_wave": {c: round(m.get(c, 0) / waves, 1) for c in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_SALU")}}
if dur:
    ns = sum(dur) / len(dur)
    out.setdefault("derived", {})["kernel_us(profiled pass)"] = round(ns * 1e-3, 2)
    if "GRBM_GUI_ACTIVE" in m:
        out["derived"]["clock_ghz_if_counter_is_summed_over_8_xcds"] = round(m["GRBM_GUI_ACTIVE"] / 8 / ns, 3)
    if "cycles_per_wave" in out["derived"]:
        out["derived"]["cycles_per_wave_over_kernel_ns"] = round(out["derived"]["cycles_per_wave"] / ns, 3)
json.dump(out, open("$O/stack_rows_pmc.json", "w"), indent=1)
print(json.dumps(out.get("derived")))
PY
grep -iE "error|invalid|not found|exceeds" $O/sp1.log $O/sp2.log $O/sp3.log | head -5
