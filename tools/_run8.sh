set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_bench.py -q -k "correlation or tail or bench or training" > gpurun_out/r8_tests.log 2>&1; echo "exit $?" >> gpurun_out/r8_tests.log
bash tools/profile_round.sh r04 > gpurun_out/r8_profile_round.log 2>&1; echo "exit $?" >> gpurun_out/r8_profile_round.log
tail -n 8 gpurun_out/r8_tests.log; tail -n 5 gpurun_out/r8_profile_round.log | cut -c1-400; ls gpurun_out/r04
