set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
python -c "import torch; print(torch.__version__, torch.cuda.get_device_name(0))" > gpurun_out/r1_env.log 2>&1
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "correlation" > gpurun_out/r1_corr_tests.log 2>&1; echo "exit $?" >> gpurun_out/r1_corr_tests.log
timeout 300 python tools/time_corr_bwd.py > gpurun_out/r1_time_bwd.log 2>&1
for a in 1 2 4 8 3; do DTT_CORR_BWD_ABLATE=$a timeout 300 python tools/time_corr_bwd.py > gpurun_out/r1_time_bwd_ablate$a.log 2>&1; done
timeout 900 python -m pytest tests/test_gpu_dist.py -x -q -s > gpurun_out/r1_dist_tests.log 2>&1; echo "exit $?" >> gpurun_out/r1_dist_tests.log
timeout 900 python bench.py --steps 10 --warmup 3 --cpu-passes 2 > gpurun_out/r1_bench.log 2>&1; echo "exit $?" >> gpurun_out/r1_bench.log
tail -3 gpurun_out/r1_corr_tests.log; cat gpurun_out/r1_time_bwd.log; tail -3 gpurun_out/r1_dist_tests.log; tail -c 1500 gpurun_out/r1_bench.log
