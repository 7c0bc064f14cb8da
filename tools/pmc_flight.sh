#!/bin/bash
# PMC passes for the flight kernel (separate rocprofv3 runs per counter group, kernel-trace only; MI355X_MICROARCH.md).
#   bash tools/pmc_flight.sh   -> gpurun_out/pmc_flight/pass*/..., summary gpurun_out/r03_pmc_flight_kernel.json
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_flight
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVES" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQC_TC_INST_REQ" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pass$i -o p -- python3 $R/bench.py --no-cpu-baseline --no-async-groups --steps 10 --warmup 3 > $OUT/pass$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/pass$i.log; exit 1; }
done
python3 - <<PY
import csv, glob, json, statistics
out = {}
for f in glob.glob("$OUT/pass*/*counter_collection.csv"):
    rows = list(csv.DictReader(open(f)))
    vals = {}
    for r in rows:
        if "flight_step_kernel" not in r.get("Kernel_Name", ""): continue
        vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for k, v in vals.items():
        v = v[2:] if len(v) > 4 else v  # drop the reset launch and the first steps
        out[k] = {"median": statistics.median(v), "min": min(v), "max": max(v), "n": len(v)}
out["_command"] = "rocprofv3 --kernel-trace --pmc <group> --output-format csv -- python3 bench.py --no-cpu-baseline --no-async-groups --steps 10 --warmup 3 (one run per group); per-launch values of flight_step_kernel at B=8192"
json.dump(out, open("$R/gpurun_out/r03_pmc_flight_kernel.json", "w"), indent=1)
for k, v in out.items():
    if isinstance(v, dict): print(k, v["median"])
PY
