#!/bin/bash
# Launch time vs batch size for both step kernels: one JSON line per run.
#   bash tools/batch_scaling.sh > gpurun_out/r01_batch_scaling.log
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for b in 2048 4096 8192 16384 32768; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-async-groups --envs-per-gpu $b --steps 200 --warmup 30 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(json.dumps({'workload': 'flight_imitation', 'envs': d['config']['envs_per_gpu'], 'ms_per_step': d['ms_per_step'], 'env_steps_per_s': d['value']}))" || exit 1
done
for b in 2048 4096 8192 16384; do
  timeout -k 10 200 python tools/bench_ball.py --batch $b --steps 60 --warmup 300 2>/dev/null | tail -1 || exit 1
done
