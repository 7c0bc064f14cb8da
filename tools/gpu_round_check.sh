# Round-end evidence on the GPU box (run through gpurun): tests, smoke, benches, rocprofv3 kernel stats, PMC passes, stamps.
#   gpurun --timeout 1200 -- 'bash tools/gpu_round_check.sh r03'
set -o pipefail
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/$TAG
O=$R/gpurun_out/$TAG
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -q -s 2>&1 | grep -v amdgpu | cut -c1-700 > $O/gpu_tests.log; tail -3 $O/gpu_tests.log
timeout -k 10 200 python __graft_entry__.py smoke 2>&1 | grep -v amdgpu | tail -3 | tee $O/smoke.log
timeout -k 10 300 python bench.py --rehearse-gather 2>&1 | grep -v amdgpu | tail -1 | tee $O/bench_flight.log | cut -c1-200
timeout -k 10 300 python bench.py --workload walk_on_ball 2>&1 | grep -v amdgpu | tail -1 | tee $O/bench_walk_on_ball.log | cut -c1-200
FLYBODY_ENV_LIB=flybody_amd/csrc/variants/libflybody_env_bstamps.so timeout -k 10 200 python tools/ball_stamps.py 2>&1 | grep -v amdgpu > $O/ball_stamp_shares.log
timeout -k 10 200 python tools/stamps.py 2>&1 | grep -v amdgpu > $O/flight_stamp_shares.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ball -o ball -- python3 $R/bench.py --workload walk_on_ball --no-cpu-baseline > $O/rocprof_ball.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_flight -o flight -- python3 $R/bench.py --no-cpu-baseline > $O/rocprof_flight.log 2>&1
cd $R && bash tools/pmc_ball.sh > $O/pmc_ball.log 2>&1; tail -2 $O/pmc_ball.log
cd $R && bash tools/pmc_flight.sh > $O/pmc_flight.log 2>&1; tail -2 $O/pmc_flight.log
cd $R && bash tools/batch_scaling.sh > $O/batch_scaling.log 2>&1; tail -3 $O/batch_scaling.log
cd $R && timeout -k 10 200 python tools/wave_timeline.py flight 2>&1 | grep -v amdgpu > $O/wave_timeline_flight.log; timeout -k 10 300 python tools/wave_timeline.py ball 2>&1 | grep -v amdgpu > $O/wave_timeline_ball.log; head -3 $O/wave_timeline_ball.log
