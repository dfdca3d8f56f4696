set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_ops.py -x -q -k "correlation" > gpurun_out/r2_corr_tests.log 2>&1; echo "exit $?" >> gpurun_out/r2_corr_tests.log
timeout 900 python -m pytest tests/test_gpu_heads.py -x -q > gpurun_out/r2_heads_tests.log 2>&1; echo "exit $?" >> gpurun_out/r2_heads_tests.log
timeout 1500 python -m pytest tests/test_gpu_model.py -x -q > gpurun_out/r2_model_tests.log 2>&1; echo "exit $?" >> gpurun_out/r2_model_tests.log
timeout 900 python -m pytest tests/test_gpu_dist.py -x -q -s > gpurun_out/r2_dist_tests.log 2>&1; echo "exit $?" >> gpurun_out/r2_dist_tests.log
(cd /tmp && ITERS=10 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_bwd -o bwd -- python $GRAFT_REPO_ROOT/tools/time_corr_bwd.py > $GRAFT_REPO_ROOT/gpurun_out/r2_time_bwd.log 2>&1)
find /tmp/prof_bwd -name "*kernel_stats*" | head -3
for f in $(find /tmp/prof_bwd -name "*kernel_stats.csv" | head -1); do head -12 $f > gpurun_out/r2_bwd_kernel_stats.csv; done
for a in 1 2 3; do (cd /tmp && DTT_CORR_BWD_ABLATE=$a ITERS=10 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_bwd_a$a -o bwd -- python $GRAFT_REPO_ROOT/tools/time_corr_bwd.py > /dev/null 2>&1); for f in $(find /tmp/prof_bwd_a$a -name "*kernel_stats.csv" | head -1); do head -6 $f > gpurun_out/r2_bwd_kernel_stats_ablate$a.csv; done; done
timeout 900 python bench.py --steps 10 --warmup 3 --cpu-passes 2 > gpurun_out/r2_bench.log 2>&1; echo "exit $?" >> gpurun_out/r2_bench.log
tail -3 gpurun_out/r2_corr_tests.log gpurun_out/r2_heads_tests.log gpurun_out/r2_model_tests.log gpurun_out/r2_dist_tests.log; cat gpurun_out/r2_time_bwd.log; cat gpurun_out/r2_bwd_kernel_stats.csv
