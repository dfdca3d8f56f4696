set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_gpu_heads.py tests/test_gpu_bench.py tests/test_gpu_model.py -q --durations=15 > gpurun_out/r7_tests.log 2>&1; echo "exit $?" >> gpurun_out/r7_tests.log
timeout 900 python bench.py --cpu-passes 3 > gpurun_out/r7_bench.log 2>&1; echo "exit $?" >> gpurun_out/r7_bench.log
DTT_PSROI_DET_FUSED=0 timeout 900 python bench.py --no-cpu-baseline --no-train-step > gpurun_out/r7_bench_unfused.log 2>&1
timeout 900 python bench.py --frames 1 --cpu-passes 3 > gpurun_out/r7_bench_frames1.log 2>&1
O=/tmp/prof
rm -rf $O; (cd /tmp && timeout 900 rocprofv3 --kernel-trace -d $O -o tr -- python $GRAFT_REPO_ROOT/bench.py --mode train --steps 5 --warmup 3 > $GRAFT_REPO_ROOT/gpurun_out/r7_prof_train_stdout.log 2>&1)
DB=$(ls $O/*.db $O/*/*.db 2>/dev/null | head -1)
python tools/rocpd_steady.py $DB 3 "corr_wsplit_kernel<3" 400 > gpurun_out/r7_train_steady_state.txt 2>&1
tail -n 30 gpurun_out/r7_tests.log; tail -c 600 gpurun_out/r7_bench.log; grep "psroi_pm_bwd\|corr_bwd\|steps=" gpurun_out/r7_train_steady_state.txt | cut -c1-170
