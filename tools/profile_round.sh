#!/bin/bash
# Regenerates the per-round evidence under gpurun_out/<round>/ on the GPU box; what is to be judged is copied to profiles/<round>_*:
#   tools/profile_round.sh r05                       every part
#   PARTS="trace calib" tools/profile_round.sh r05   some parts (bench trace pmc cfg traintrace bwdpmc bwdabl micro calib)
# Every file it writes is listed, with the command that made it and the figures read from it, in <round>/README_rows.md by
# tools/profiles_readme.py -- profiles/README.md's table of the round is that file, not typed by hand.
R=${1:-r05}
PARTS=${PARTS:-bench trace pmc cfg traintrace bwdpmc micro calib}
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}; export TMPDIR=/tmp
O=$PWD/gpurun_out/$R; mkdir -p $O
has() { [[ " $PARTS " == *" $1 "* ]]; }
PMC="FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
firstdb() { ls $1/*.db $1/*/*.db 2>/dev/null | head -1; }

if has bench; then   # the driver's own command: default flags, stdout only (ONE line)
  timeout 900 python bench.py > $O/bench_stdout.log 2> $O/bench_stderr.log; tail -c 400 $O/bench_stderr.log
fi

if has trace; then   # the inference step under rocprofv3: INFERENCE STEPS ONLY (--no-train-step: the training leg launches the same kernels)
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace -o bench -- python bench.py --no-cpu-baseline --no-train-step --steps 10 --warmup 5 > $O/prof_bench_stdout.log 2>&1
  DB=$(firstdb $O/trace)
  python tools/rocpd_sequence.py $DB "psroi_pm_det_kernel" > $O/bench_step_sequence.txt 2>&1
  N=$(wc -l < $O/bench_step_sequence.txt)                                   # launches of ONE step: what every steady-state step must hold
  python tools/rocpd_stats.py $DB > $O/bench_kernel_stats.txt 2>&1
  python tools/rocpd_steady.py $DB 5 "psroi_pm_det_kernel" 40 --expect $N > $O/bench_steady_state.txt 2>&1 || echo "profile_round: steady-state window REFUSED"
  python tools/rocpd_tail_steps.py $DB 8 > $O/bench_tail_overlap.txt 2>&1 || echo "profile_round: tail window REFUSED"
  rm -rf $O/trace
fi

if has pmc; then     # counters of the tail's hand-written kernels, ONE counter per pass (no trace domains beside --kernel-trace)
  rm -f $O/pmc_tail.txt
  for c in $PMC; do
    timeout 400 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_$c -o p -- python tools/pmc_tail.py > $O/pmc_$c.log 2>&1
    python tools/rocpd_pmc.py $(firstdb $O/pmc_$c) | grep -v "at::\|Cijk\|miopen\|elementwise\|rocclr\|rocprim" >> $O/pmc_tail.txt 2>&1
    rm -f $O/pmc_$c.log
  done
  # the traffic figure bench.py quotes: regenerated on THIS binary (the json carries the library's sha256)
  python tools/pmc_conv5_json.py $(firstdb $O/pmc_FETCH_SIZE) $(firstdb $O/pmc_WRITE_SIZE) $O/pmc_conv5.json > $O/pmc_conv5.log 2>&1 \
    || { echo "profile_round: pmc_conv5.json not produced"; cat $O/pmc_conv5.log; }
  for c in $PMC; do rm -rf $O/pmc_$c; done
fi

if has bwdpmc; then  # counters of the streamed gradient kernels, one map per pass (per-kernel averages then belong to it) + the json bench.py quotes
  rm -f $O/pmc_corr_bwd.txt; ARGS=""
  for m in conv5 conv4 conv3; do
    for c in $PMC; do
      (cd /tmp && ONLY=$m ITERS=5 timeout 400 rocprofv3 --kernel-trace --pmc $c -d $O/bw_${m}_$c -o p -- python $OLDPWD/tools/time_corr_bwd.py > $O/tp.log 2>&1)
      echo "== $m  $c" >> $O/pmc_corr_bwd.txt
      python tools/rocpd_pmc.py $(firstdb $O/bw_${m}_$c) 2>&1 | grep "corr_bwd_stream\|corr_bwd_band" >> $O/pmc_corr_bwd.txt
      rm -f $O/tp.log
    done
    ARGS="$ARGS $m:$(firstdb $O/bw_${m}_FETCH_SIZE):$(firstdb $O/bw_${m}_WRITE_SIZE)"
  done
  for m in conv5 conv4 conv3; do   # BASELINE configs[4]'s per-rank shape: B = 1, 563 x 1000, d = 16 (window radius 16: the quarters inside ONE launch)
    for c in FETCH_SIZE WRITE_SIZE; do
      (cd /tmp && ONLY=$m B=1 D=16 SHAPE=563 NO_OLD=1 ITERS=5 timeout 400 rocprofv3 --kernel-trace --pmc $c -d $O/bw16_${m}_$c -o p -- python $OLDPWD/tools/time_corr_bwd.py > $O/tp.log 2>&1)
      echo "== d16 $m  $c" >> $O/pmc_corr_bwd.txt
      python tools/rocpd_pmc.py $(firstdb $O/bw16_${m}_$c) 2>&1 | grep "corr_bwd_stream\|corr_bwd_band" >> $O/pmc_corr_bwd.txt
      rm -f $O/tp.log
    done
    ARGS="$ARGS d16_$m:$(firstdb $O/bw16_${m}_FETCH_SIZE):$(firstdb $O/bw16_${m}_WRITE_SIZE)"
  done
  python tools/pmc_corr_bwd_json.py $O/pmc_corr_bwd.json $ARGS > $O/pmc_corr_bwd.log 2>&1 || { echo "profile_round: pmc_corr_bwd.json not produced"; cat $O/pmc_corr_bwd.log; }
  rm -rf $O/bw_conv?_* $O/bw16_conv?_*
  # the PSRoI backward launch (class + box heads of the training step's 10184 pixels; the tracking head's): bytes fetched / written per launch
  rm -f $O/pmc_psroi_bwd.txt
  for c in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c -d $O/pb_$c -o p -- python $OLDPWD/tools/time_psroi_bwd.py --iters 20 > $O/tp.log 2>&1)
    python tools/rocpd_pmc.py $(firstdb $O/pb_$c) 2>&1 | grep "psroi_pm_bwd" >> $O/pmc_psroi_bwd.txt
    rm -rf $O/pb_$c $O/tp.log
  done
  timeout 300 python tools/time_psroi_bwd.py > $O/psroi_bwd_microbench.txt 2>&1
fi

if has cfg; then     # the other BASELINE configurations; configs[4] with its per-rank TRAINING step riding along (secondary.train_step)
  timeout 900 python bench.py --no-cpu-baseline --pooling align --disp 16 --height 563 --width 1000 --batch 1 > $O/bench_config4_stdout.log 2>/dev/null
  timeout 900 python bench.py --frames 1 --no-train-step > $O/bench_frames1_stdout.log 2>/dev/null     # BASELINE configs[1]: single-frame R-FCN
  timeout 900 python bench.py --mode train --steps 8 --warmup 4 > $O/bench_train_stdout.log 2>/dev/null
  timeout 900 python bench.py --mode train --steps 8 --warmup 4 --pooling align --disp 16 --height 563 --width 1000 --batch 1 > $O/bench_train_config4_stdout.log 2>/dev/null
fi

if has traintrace; then
  timeout 600 rocprofv3 --kernel-trace -d $O/trace -o tr -- python bench.py --mode train --steps 5 --warmup 3 > $O/prof_train_stdout.log 2>&1
  python tools/rocpd_steady.py $(firstdb $O/trace) 3 "psroi_pm_bwd_rows_kernel<7, 16, 2>" 400 > $O/train_steady_state.txt 2>&1 || echo "profile_round: training window REFUSED"
  python tools/rocpd_stats.py $(firstdb $O/trace) > $O/train_kernel_stats.txt 2>&1
  rm -rf $O/trace
fi

if has bwdabl; then  # the correlation gradient kernels alone with the DMA / the MFMAs / both ablated (DTT_CORR_BWD_ABLATE 1 / 2 / 3)
  rm -f $O/corr_bwd_ablation.txt
  for a in 0 1 2 3; do
    (cd /tmp && DTT_CORR_BWD_ABLATE=$a ITERS=10 timeout 600 rocprofv3 --kernel-trace -d $O/trace -o bwd -- python $OLDPWD/tools/time_corr_bwd.py > $O/tb$a.log 2>&1)
    echo "== DTT_CORR_BWD_ABLATE=$a  (0 = the shipped kernel; 1 no LDS-DMA, 2 no operand reads / MFMAs, 3 neither)" >> $O/corr_bwd_ablation.txt
    python tools/rocpd_stats.py $(firstdb $O/trace) 2>&1 | grep -i "corr_bwd\|kernel " >> $O/corr_bwd_ablation.txt
    rm -rf $O/trace
  done
  grep "gradients\|diff" $O/tb0.log >> $O/corr_bwd_ablation.txt; rm -f $O/tb?.log
fi

if has micro; then   # the proposal layer and the forward correlations alone
  rm -f $O/proposal_microbench.txt
  for b in 2 4; do echo "== B=$b" >> $O/proposal_microbench.txt; B=$b timeout 300 python tools/time_proposal.py 2>&1 | grep -v "Warn\|amdgpu.ids\|capture_end" >> $O/proposal_microbench.txt; done
  B=4 TEST_ONLY=1 timeout 300 rocprofv3 --kernel-trace -d $O/trace -o prop -- python tools/time_proposal.py > /dev/null 2>&1
  python tools/rocpd_stats.py $(firstdb $O/trace) | head -8 >> $O/proposal_microbench.txt 2>&1
  rm -rf $O/trace
  ITERS=20 timeout 300 python tools/time_corr.py 2>&1 | grep -v "Warn\|amdgpu.ids" > $O/corr_microbench.txt
  B=8 ITERS=10 timeout 300 python tools/time_corr.py 2>&1 | grep -v "Warn\|amdgpu.ids" >> $O/corr_microbench.txt
fi

if has calib; then   # what FETCH_SIZE / WRITE_SIZE count on the kernels' own access patterns (binaries: hipcc -O3 tools/probes/*.hip -> tools/_variants/)
  if [ -x tools/_variants/fetch_calib ]; then
    timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/calib -o p -- tools/_variants/fetch_calib > $O/fetch_calib.log 2>&1
    python tools/rocpd_pmc.py $(firstdb $O/calib) > $O/fetch_calib.txt; tail -1 $O/fetch_calib.log >> $O/fetch_calib.txt; rm -rf $O/calib $O/fetch_calib.log
  fi
  if [ -x tools/_variants/write_calib ]; then
    timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/calib -o p -- tools/_variants/write_calib > $O/write_calib.log 2>&1
    python tools/rocpd_pmc.py $(firstdb $O/calib) > $O/write_calib.txt; tail -1 $O/write_calib.log >> $O/write_calib.txt; rm -rf $O/calib $O/write_calib.log
  fi
fi
python tools/profiles_readme.py $R $O > $O/README_rows.md 2> $O/README_rows.err || cat $O/README_rows.err
[ -f $O/bench_stdout.log ] && tail -c 1500 $O/bench_stdout.log
exit 0
