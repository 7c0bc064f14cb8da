# Round-end evidence, part C (PMC passes of the walk_on_ball kernel, batch scaling, overflow rates at full-range actions).
set -o pipefail
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/$TAG
O=$R/gpurun_out/$TAG
cd $R && bash tools/pmc_ball.sh > $O/pmc_ball.log 2>&1; tail -2 $O/pmc_ball.log
cd $R && bash tools/batch_scaling.sh > $O/batch_scaling.log 2>&1; tail -3 $O/batch_scaling.log
cd $R && timeout -k 10 300 python tools/ball_overflow_stats.py 2>&1 | grep -v amdgpu > $O/ball_overflow_rates.log; tail -3 $O/ball_overflow_rates.log
cd $R && timeout -k 10 400 python tools/soak.py 2>&1 | grep -v amdgpu > $O/soak.log; cat $O/soak.log | cut -c1-300
cd $R && timeout -k 10 300 python tools/bench_actor_loop.py 2>&1 | grep -v amdgpu > $O/actor_loop_throughput.log; tail -4 $O/actor_loop_throughput.log | cut -c1-300
