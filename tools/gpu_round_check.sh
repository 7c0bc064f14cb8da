set -o pipefail
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd $R
timeout -k 10 400 python -m pytest tests -m gpu -q -s 2>&1 | grep -v amdgpu | tail -40 | cut -c1-600 > gpurun_out/r01b_gpu_tests.log; tail -25 gpurun_out/r01b_gpu_tests.log
timeout -k 10 200 python __graft_entry__.py smoke 2>&1 | grep -v amdgpu | tail -3 | tee gpurun_out/r01b_smoke.log
timeout -k 10 300 python bench.py --workload walk_on_ball 2>&1 | grep -v amdgpu | tail -1 | tee gpurun_out/r01b_bench_walk_on_ball.log
timeout -k 10 300 python bench.py 2>&1 | grep -v amdgpu | tail -1 | tee gpurun_out/r01b_bench_default_run.log
FLYBODY_ENV_LIB=flybody_amd/csrc/variants/libflybody_env_bstamps.so timeout -k 10 200 python tools/ball_stamps.py 2>&1 | grep -v amdgpu > gpurun_out/r01b_ball_stamp_shares.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ball -o ball -- python3 $R/bench.py --workload walk_on_ball --no-cpu-baseline > $R/gpurun_out/r01b_rocprof_ball.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_flight -o flight -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/r01b_rocprof_flight.log 2>&1
ls -R $R/gpurun_out/prof_ball $R/gpurun_out/prof_flight | head -20
cd $R && bash tools/pmc_ball.sh > gpurun_out/r01b_pmc_ball.log 2>&1; tail -3 gpurun_out/r01b_pmc_ball.log
