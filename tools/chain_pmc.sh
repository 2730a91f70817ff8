#!/bin/bash
# PMC passes on the chained layer-1 kernel (l1_chain.hip) alone (--separate-tail: its trailing Adam-tail workgroups run as
# their own launch, so the kernel is exactly 256 workgroups x 8 waves), on the GPU box from the repo root:
#   bash tools/chain_pmc.sh   ->  gpurun_out/chain_pmc.json
# Separate rocprofv3 runs per counter set, --kernel-trace only beside --pmc; every run under `timeout`.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
rm -rf $O/cp1 $O/cp2 $O/cp3 $O/cp4
B="python3 $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-l1-gemm --no-graph --separate-tail"
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES -d $O/cp1 -o p --output-format csv -- $B > $O/cp1.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU -d $O/cp2 -o p --output-format csv -- $B > $O/cp2.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS -d $O/cp3 -o p --output-format csv -- $B > $O/cp3.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $O/cp4 -o p --output-format csv -- $B > $O/cp4.log 2>&1
python3 - <<PY
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
short = lambda n: n.split("(")[0].replace("void ", "")
want = lambda k: "l1_bwd_adam_chain" in k
for d in ("cp1", "cp2", "cp3", "cp4"):
    for f in glob.glob("$O/" + d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if want(k):
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob("$O/cp4/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if want(k):
            dur[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
out = {"source": "tools/chain_pmc.sh (bench.py --steps 2 --warmup 2 --no-graph --separate-tail; 1000 x 100,000 SNPs)"}
for k, v in acc.items():
    m = {c: sum(x) / len(x) for c, x in v.items()}
    waves = 2048.0
    if "SQ_WAVE_CYCLES" in m:
        cyc = 4 * m["SQ_WAVE_CYCLES"] / waves
        f = lambda c: round(4 * m.get(c, 0) / (waves * cyc), 3)
        m["derived"] = {"cycles_per_wave": round(cyc), "mfma_busy": round(m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * cyc), 3),
                        "wait_any(parked)": f("SQ_WAIT_ANY"), "wait_inst_any(issue stall)": f("SQ_WAIT_INST_ANY"),
                        "active_inst_any": f("SQ_ACTIVE_INST_ANY"), "valu": f("SQ_ACTIVE_INST_VALU"), "lds_inst": f("SQ_ACTIVE_INST_LDS"),
                        "vmem_inst": f("SQ_ACTIVE_INST_VMEM"), "sca": f("SQ_ACTIVE_INST_SCA"), "misc": f("SQ_ACTIVE_INST_MISC"),
                        "wait_inst_lds": f("SQ_WAIT_INST_LDS"),
                        "lds_array_busy": round(m.get("SQ_LDS_IDX_ACTIVE", 0) / (256 * cyc), 3),
                        "lds_conflict": round(m.get("SQ_LDS_BANK_CONFLICT", 0) / max(m.get("SQ_LDS_IDX_ACTIVE", 1), 1), 3),
                        "insts_per_wave": {c: round(m.get(c, 0) / waves, 1) for c in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_SALU")},
                        "level_vmem(avg outstanding x cycles)": m.get("SQ_INST_LEVEL_VMEM")}
    if k in dur and "GRBM_GUI_ACTIVE" in m:
        ns = sum(dur[k]) / len(dur[k])
        m.setdefault("derived", {})["kernel_us(profiled pass)"] = round(ns * 1e-3, 2)
        m["derived"]["clock_ghz_if_counter_is_summed_over_8_xcds"] = round(m["GRBM_GUI_ACTIVE"] / 8 / ns, 3)
        if "cycles_per_wave" in m["derived"]:
            m["derived"]["clock_ghz_from_wave_cycles"] = round(m["derived"]["cycles_per_wave"] / ns, 3)
    out[k] = m
json.dump(out, open("$O/chain_pmc.json", "w"), indent=1)
for k, m in out.items():
    if isinstance(m, dict):
        print(k, json.dumps(m.get("derived")))
PY
grep -iE "error|invalid|not found|exceeds" $O/cp1.log $O/cp2.log $O/cp3.log $O/cp4.log | head -5
