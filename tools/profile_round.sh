#!/bin/bash
# Regenerates the per-round evidence under gpurun_out/<round>/ on the GPU box (copy what is to be judged into profiles/):
#   tools/profile_round.sh r02
# bench stdout (default flags), rocprofv3 kernel-trace summaries (whole run, steady state, one step launch by launch, the tail's
# per-step overlap), separate --pmc passes over the tail's hand-written kernels, the config-4 / config-3 lines.
R=${1:-r04}
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}; export TMPDIR=/tmp
O=$PWD/gpurun_out/$R; rm -rf $O; mkdir -p $O
timeout 900 python bench.py > $O/bench_stdout.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace -o bench -- python bench.py --no-cpu-baseline --steps 10 --warmup 5 > $O/prof_bench_stdout.log 2>&1
DB=$(ls $O/trace/*.db | head -1)
python tools/rocpd_stats.py $DB > $O/bench_kernel_stats.txt 2>&1
python tools/rocpd_steady.py $DB 5 "corr_wsplit_kernel<9" > $O/bench_steady_state.txt 2>&1
python tools/rocpd_sequence.py $DB "psroi_pm_det_kernel" > $O/bench_step_sequence.txt 2>&1
python tools/rocpd_tail_steps.py $DB 8 > $O/bench_tail_overlap.txt 2>&1
rm -rf $O/trace
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE; do
  timeout 400 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_$c -o p -- python tools/pmc_tail.py > $O/pmc_$c.log 2>&1
  python tools/rocpd_pmc.py $(ls $O/pmc_$c/*.db | head -1) | grep -v "at::\|Cijk\|miopen\|elementwise\|rocclr\|rocprim" >> $O/pmc_tail.txt 2>&1
  rm -f $O/pmc_$c.log
done
# the traffic figure bench.py quotes: regenerated on THIS binary (the json carries the library's sha256)
python tools/pmc_conv5_json.py $(ls $O/pmc_FETCH_SIZE/*.db | head -1) $(ls $O/pmc_WRITE_SIZE/*.db | head -1) $O/pmc_conv5.json > $O/pmc_conv5.log 2>&1 \
  || { echo "profile_round: pmc_conv5.json not produced"; cat $O/pmc_conv5.log; exit 1; }
[ $O/pmc_conv5.json -nt pytorch-detect-to-track_amd/lib/libdtt_hip.so ] || { echo "profile_round: pmc_conv5.json is older than libdtt_hip.so"; exit 1; }
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE; do rm -rf $O/pmc_$c; done
timeout 600 python bench.py --no-cpu-baseline --no-train-step --pooling align --disp 16 --height 563 --width 1000 --batch 1 > $O/bench_config4_stdout.log 2>&1
timeout 900 python bench.py --frames 1 --no-train-step > $O/bench_frames1_stdout.log 2>&1     # BASELINE configs[1]: single-frame R-FCN
timeout 900 python bench.py --mode train --steps 8 --warmup 4 > $O/bench_train_stdout.log 2>&1
timeout 600 rocprofv3 --kernel-trace -d $O/trace -o tr -- python bench.py --mode train --steps 5 --warmup 3 > $O/prof_train_stdout.log 2>&1
python tools/rocpd_steady.py $(ls $O/trace/*.db | head -1) 3 "corr_wsplit_kernel<3" 400 > $O/train_steady_state.txt 2>&1
python tools/rocpd_stats.py $(ls $O/trace/*.db | head -1) > $O/train_kernel_stats.txt 2>&1
rm -rf $O/trace
# counters of the streamed gradient kernels, one map per pass (per-kernel averages then belong to it)
for m in conv5 conv4; do
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE SQ_WAVE_CYCLES; do
  (cd /tmp && ONLY=$m ITERS=5 timeout 400 rocprofv3 --kernel-trace --pmc $c -d $O/trace -o p -- python $OLDPWD/tools/time_corr_bwd.py > $O/tp.log 2>&1)
  echo "== $m  $c" >> $O/pmc_corr_bwd.txt
  python tools/rocpd_pmc.py $(ls $O/trace/*.db $O/trace/*/*.db 2>/dev/null | head -1) 2>&1 | grep "corr_bwd_stream\|corr_bwd_band" >> $O/pmc_corr_bwd.txt
  rm -rf $O/trace $O/tp.log
done
done
# the correlation gradient kernels alone (conv5 / conv4 / conv3 at B = 2: streamed kernels next to round 1's), per kernel by rocprofv3,
# and with the DMA / the MFMAs / both ablated (DTT_CORR_BWD_ABLATE 1 / 2 / 3)
for a in 0 1 2 3; do
  (cd /tmp && DTT_CORR_BWD_ABLATE=$a ITERS=10 timeout 600 rocprofv3 --kernel-trace -d $O/trace -o bwd -- python $OLDPWD/tools/time_corr_bwd.py > $O/tb$a.log 2>&1)
  echo "== DTT_CORR_BWD_ABLATE=$a  (0 = the shipped kernel; 1 no LDS-DMA, 2 no operand reads / MFMAs, 3 neither)" >> $O/corr_bwd_ablation.txt
  python tools/rocpd_stats.py $(ls $O/trace/*.db $O/trace/*/*.db 2>/dev/null | head -1) 2>&1 | grep -i "corr_bwd\|kernel " >> $O/corr_bwd_ablation.txt
  rm -rf $O/trace
done
grep "gradients\|diff" $O/tb0.log >> $O/corr_bwd_ablation.txt; rm -f $O/tb?.log
# the proposal layer alone (B = 2: one frame pair, the bench step; B = 4: two pairs), per-kernel split of the B = 4 call
for b in 2 4; do echo "== B=$b" >> $O/proposal_microbench.txt; B=$b timeout 300 python tools/time_proposal.py 2>&1 | grep -v "Warn\|amdgpu.ids\|capture_end" >> $O/proposal_microbench.txt; done
B=4 TEST_ONLY=1 timeout 300 rocprofv3 --kernel-trace -d $O/trace -o prop -- python tools/time_proposal.py > /dev/null 2>&1
python tools/rocpd_stats.py $(ls $O/trace/*.db | head -1) | head -6 >> $O/proposal_microbench.txt 2>&1
rm -rf $O/trace
ITERS=20 timeout 300 python tools/time_corr.py > $O/corr_microbench.txt 2>&1
B=8 ITERS=10 timeout 300 python tools/time_corr.py >> $O/corr_microbench.txt 2>&1
# hardware probes / placement traces behind DESIGN.md 4.10 (binaries: hipcc -O3 tools/probes/*.hip -> tools/_variants/; tools/build_trace_variant.sh)
[ -x tools/_variants/wg_placement ] && timeout 120 tools/_variants/wg_placement > $O/wg_placement_probe.txt 2>&1
if [ -x tools/_variants/fetch_calib ]; then
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/calib -o p -- tools/_variants/fetch_calib > $O/fetch_calib.log 2>&1
  python tools/rocpd_pmc.py $O/calib/*.db > $O/fetch_calib.txt; tail -1 $O/fetch_calib.log >> $O/fetch_calib.txt; rm -rf $O/calib $O/fetch_calib.log
fi
if [ -x tools/_variants/write_calib ]; then
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/calib -o p -- tools/_variants/write_calib > $O/write_calib.log 2>&1
  python tools/rocpd_pmc.py $O/calib/*.db > $O/write_calib.txt; tail -1 $O/write_calib.log >> $O/write_calib.txt; rm -rf $O/calib $O/write_calib.log
fi
[ -f tools/_variants/wgtrace.so ] && DTT_HIP_LIBRARY=$PWD/tools/_variants/wgtrace.so timeout 600 python tools/wg_trace.py 2>&1 | grep -v "Warn\|amdgpu.ids" > $O/wg_trace.txt
[ -f tools/_variants/wstrace.so ] && DTT_HIP_LIBRARY=$PWD/tools/_variants/wstrace.so timeout 300 python tools/ws_trace.py 2>&1 | grep -v "Warn\|amdgpu.ids" > $O/ws_trace.txt
tail -c 1500 $O/bench_stdout.log
