#!/bin/bash
# round-5 GPU call A: new tests, repaired trace pipeline, FETCH calibration, head GEMM store-form A/B under --pmc WRITE_SIZE
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp; O=$PWD/gpurun_out/r5a; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_train_fullsize.py -x -q -s > $O/tests_train_fullsize.log 2>&1; echo "train_fullsize rc=$?"
timeout 1700 python -m pytest tests/test_gpu_bench.py -x -q > $O/tests_bench.log 2>&1; echo "bench tests rc=$?"
PARTS="trace calib" bash tools/profile_round.sh r5a_prof > $O/profile_round.log 2>&1
for v in base v1 v2; do
  L=""; [ $v != base ] && L=$PWD/tools/_variants/heads_$v.so
  for P in 0 3; do
    (cd /tmp && DTT_HIP_LIBRARY=$L PASSES=$P ITERS=5 timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/hw -o p -- python $OLDPWD/tools/time_head_gemm.py > $O/hw.log 2>&1)
    echo "== $v PASSES=$P WRITE_SIZE" >> $O/head_store.txt; python tools/rocpd_pmc.py $(ls $O/hw/*.db $O/hw/*/*.db 2>/dev/null | head -1) | grep head_gemm >> $O/head_store.txt; rm -rf $O/hw
    (cd /tmp && DTT_HIP_LIBRARY=$L PASSES=$P ITERS=20 timeout 300 rocprofv3 --kernel-trace -d $O/hw -o p -- python $OLDPWD/tools/time_head_gemm.py > $O/hw.log 2>&1)
    echo "== $v PASSES=$P time" >> $O/head_store.txt; grep "head_gemm cls" $O/hw.log >> $O/head_store.txt; python tools/rocpd_stats.py $(ls $O/hw/*.db $O/hw/*/*.db 2>/dev/null | head -1) | grep head_gemm >> $O/head_store.txt; rm -rf $O/hw
  done
done
DTT_HIP_LIBRARY=$PWD/tools/_variants/heads_v1.so timeout 900 python -m pytest tests/test_gpu_heads.py -x -q > $O/tests_heads_v1.log 2>&1; echo "heads v1 rc=$?"
tail -3 $O/tests_train_fullsize.log; tail -3 $O/tests_bench.log; cat $O/head_store.txt | cut -c1-200
