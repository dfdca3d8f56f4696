set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/ -q -m gpu > gpurun_out/r6_gpu_tests.log 2>&1; echo "exit $?" >> gpurun_out/r6_gpu_tests.log
timeout 900 python bench.py > gpurun_out/r6_bench.log 2>&1; echo "exit $?" >> gpurun_out/r6_bench.log
O=/tmp/prof
rm -rf $O; (cd /tmp && timeout 900 rocprofv3 --kernel-trace -d $O -o tr -- python $GRAFT_REPO_ROOT/bench.py --mode train --steps 5 --warmup 3 > $GRAFT_REPO_ROOT/gpurun_out/r6_prof_train_stdout.log 2>&1)
DB=$(ls $O/*.db $O/*/*.db 2>/dev/null | head -1)
python tools/rocpd_steady.py $DB 3 "corr_wsplit_kernel<3" 400 > gpurun_out/r6_train_steady_state.txt 2>&1
tail -n 15 gpurun_out/r6_gpu_tests.log; tail -c 800 gpurun_out/r6_bench.log; grep "psroi_pm_bwd\|corr_bwd\|steps=" gpurun_out/r6_train_steady_state.txt | cut -c1-170
