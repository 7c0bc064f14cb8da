# Round-end evidence, part A (tests, smoke, the two bench lines).   gpurun --timeout 1200 -- 'bash tools/gpu_round_check_a.sh r03'
set -o pipefail
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/$TAG
O=$R/gpurun_out/$TAG
cd $R
timeout -k 10 800 python -m pytest tests -m gpu -q -s 2>&1 | grep -v amdgpu | cut -c1-900 > $O/gpu_tests.log; tail -3 $O/gpu_tests.log
timeout -k 10 200 python __graft_entry__.py smoke 2>&1 | grep -v amdgpu | tail -3 | tee $O/smoke.log
timeout -k 10 300 python bench.py --rehearse-gather 2>&1 | grep -v amdgpu | tail -1 | tee $O/bench_flight.log | cut -c1-200
timeout -k 10 300 python bench.py --workload walk_on_ball 2>&1 | grep -v amdgpu | tail -1 | tee $O/bench_walk_on_ball.log | cut -c1-200
timeout -k 10 300 python bench.py --workload walk_on_ball --action-amplitude 1.0 --no-cpu-baseline 2>&1 | grep -v amdgpu | tail -1 | tee $O/bench_walk_on_ball_amp1.log | cut -c1-200
