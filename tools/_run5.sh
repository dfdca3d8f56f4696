set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "correlation" > gpurun_out/r5_corr_tests.log 2>&1; echo "exit $?" >> gpurun_out/r5_corr_tests.log
timeout 900 python -m pytest tests/test_gpu_heads.py -x -q > gpurun_out/r5_heads_tests.log 2>&1; echo "exit $?" >> gpurun_out/r5_heads_tests.log
O=/tmp/prof
for a in 0 2 3; do
  rm -rf $O; (cd /tmp && DTT_CORR_BWD_ABLATE=$a ITERS=10 timeout 600 rocprofv3 --kernel-trace -d $O -o bwd -- python $GRAFT_REPO_ROOT/tools/time_corr_bwd.py > /tmp/tb$a.log 2>&1)
  echo "== ablate $a" >> gpurun_out/r5_bwd_kernel_stats.txt
  python tools/rocpd_stats.py $(ls $O/*.db $O/*/*.db 2>/dev/null | head -1) 2>&1 | grep -i "corr_bwd\|kernel " >> gpurun_out/r5_bwd_kernel_stats.txt
done
grep "diff\|gradients" /tmp/tb0.log >> gpurun_out/r5_bwd_kernel_stats.txt
timeout 1500 python -m pytest tests/test_gpu_model.py -x -q > gpurun_out/r5_model_tests.log 2>&1; echo "exit $?" >> gpurun_out/r5_model_tests.log
timeout 900 python bench.py --mode train --steps 8 --warmup 4 > gpurun_out/r5_bench_train.log 2>&1
DTT_TRAIN_PM=0 timeout 900 python bench.py --mode train --steps 8 --warmup 4 > gpurun_out/r5_bench_train_nopm.log 2>&1
rm -rf $O; (cd /tmp && timeout 900 rocprofv3 --kernel-trace -d $O -o tr -- python $GRAFT_REPO_ROOT/bench.py --mode train --steps 5 --warmup 3 > $GRAFT_REPO_ROOT/gpurun_out/r5_prof_train_stdout.log 2>&1)
DB=$(ls $O/*.db $O/*/*.db 2>/dev/null | head -1)
python tools/rocpd_steady.py $DB 3 "corr_wsplit_kernel<3" 400 > gpurun_out/r5_train_steady_state.txt 2>&1
tail -n 3 gpurun_out/r5_corr_tests.log gpurun_out/r5_heads_tests.log gpurun_out/r5_model_tests.log; cat gpurun_out/r5_bwd_kernel_stats.txt; grep -o '"ms_per_step": [0-9.]*' gpurun_out/r5_bench_train*.log; grep "psroi_pm_bwd\|corr_bwd\|steps=" gpurun_out/r5_train_steady_state.txt | cut -c1-170
