#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the flight kernel only (two rocprofv3 passes), printed as MB per launch.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_traffic
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -o p -- python3 $R/bench.py --no-cpu-baseline --no-async-groups --steps 10 --warmup 3 > $OUT/$c.log 2>&1 || { echo "$c failed"; tail -3 $OUT/$c.log; }
done
python3 - <<PY
import csv, glob, statistics
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    v = []
    for f in glob.glob("$OUT/%s/*counter_collection.csv" % c):
        for r in csv.DictReader(open(f)):
            if "flight_step_kernel" in r.get("Kernel_Name", "") and r["Counter_Name"] == c: v.append(float(r["Counter_Value"]))
    v = v[2:] if len(v) > 4 else v
    print(c, "median %.1f MB per launch" % (statistics.median(v) * 1024 / 1e6) if v else "none")
PY
