#!/bin/bash
# The correlation GRADIENT ops of BASELINE configs[4] (563x1000 frames, d = 16, one pair per rank) alone: per-kernel times
# (rocprofv3 --kernel-trace) and HBM traffic (separate --pmc FETCH_SIZE / WRITE_SIZE passes) -> gpurun_out/<tag>/bwd16_*.txt
#   tools/bwd16_probe.sh <tag> [B]
T=${1:-bwd16}; B=${2:-1}
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}; export TMPDIR=/tmp
O=$PWD/gpurun_out/$T; mkdir -p $O
firstdb() { ls $1/*.db $1/*/*.db 2>/dev/null | head -1; }
RUN="env B=$B D=16 SHAPE=563 NO_OLD=1 ITERS=10 python $PWD/tools/time_corr_bwd.py"
(cd /tmp && timeout 300 $RUN) 2>&1 | grep -v "Warn\|amdgpu.ids" > $O/bwd16_times.txt
(cd /tmp && timeout 400 rocprofv3 --kernel-trace -d $O/tr -o t -- $RUN > /dev/null 2>&1)
python tools/rocpd_stats.py $(firstdb $O/tr) 2>&1 | grep -i "corr_bwd\|kernel " >> $O/bwd16_times.txt; rm -rf $O/tr
rm -f $O/bwd16_pmc.txt
for m in conv5 conv4 conv3; do
  for c in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && ONLY=$m timeout 400 rocprofv3 --kernel-trace --pmc $c -d $O/p_${m}_$c -o p -- $RUN > /dev/null 2>&1)
    echo "== $m  $c" >> $O/bwd16_pmc.txt
    python tools/rocpd_pmc.py $(firstdb $O/p_${m}_$c) 2>&1 | grep "corr_bwd_stream\|corr_bwd_band\|kernel" >> $O/bwd16_pmc.txt
    [ $m$c = conv3WRITE_SIZE ] && python -c "import sqlite3,sys; db=sqlite3.connect(sys.argv[1]); print([r[1] for r in db.execute('pragma table_info(counters_collection)')]); print(db.execute('select * from counters_collection limit 1').fetchall())" $(firstdb $O/p_${m}_$c) > $O/pmc_columns.txt 2>&1
    rm -rf $O/p_${m}_$c
  done
done
cat $O/bwd16_times.txt $O/bwd16_pmc.txt
