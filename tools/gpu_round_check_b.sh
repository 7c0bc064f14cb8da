# Round-end evidence, part B (stamps, rocprofv3 kernel stats, PMC passes of the flight kernel).
set -o pipefail
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/$TAG
O=$R/gpurun_out/$TAG
cd $R
FLYBODY_ENV_LIB=flybody_amd/csrc/variants/libflybody_env_bstamps.so timeout -k 10 200 python tools/ball_stamps.py 2>&1 | grep -v amdgpu > $O/ball_stamp_shares.log
timeout -k 10 200 python tools/stamps.py 2>&1 | grep -v amdgpu > $O/flight_stamp_shares.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ball -o ball -- python3 $R/bench.py --workload walk_on_ball --no-cpu-baseline --no-async-groups > $O/rocprof_ball.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_flight -o flight -- python3 $R/bench.py --no-cpu-baseline --no-async-groups > $O/rocprof_flight.log 2>&1
cd $R && bash tools/pmc_flight.sh > $O/pmc_flight.log 2>&1; tail -2 $O/pmc_flight.log
