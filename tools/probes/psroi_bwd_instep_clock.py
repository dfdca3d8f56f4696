#!/usr/bin/env python3
"""Shader clock and workgroup lifetimes of the LAST psroi_pm_bwd_rows_kernel launch of a `bench.py --mode train` run (developer tool; the
DTT_PSROI_BWD_STAMP build): DTT_HIP_LIBRARY=tools/_variants/pbstamp.so python tools/probes/psroi_bwd_instep_clock.py --mode train --steps 6 --warmup 3"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-detect-to-track_amd")]
import numpy as np
import bench
bench.main()
from dtt import _lib
L = _lib.lib()
buf = (ctypes.c_ulonglong * 256)(); wgb = (ctypes.c_ulonglong * (1024 * 3))()
L.dtt_psroi_bwd_stamps_read.argtypes = [ctypes.c_void_p, ctypes.c_int]; L.dtt_psroi_bwd_wg_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert L.dtt_psroi_bwd_stamps_read(buf, 256) and L.dtt_psroi_bwd_wg_read(wgb, 1024 * 3)
st = np.array(buf, dtype=np.uint64).reshape(4, 64).astype(np.int64)
for row in (0, 2):
    ticks, real = st[row, 50] - st[row, 0], st[row, 63] - st[row, 62]
    print("last launch: workgroup %s: %d shader-clock ticks in %.2f us -> %.0f MHz; prologue (to the first pixel) %d ticks" % (
        "0" if row == 0 else "middle", ticks, real / 100.0, ticks / max(real, 1) * 100.0, st[row, 4] - st[row, 0]))
w_all = np.array(wgb, dtype=np.uint64).reshape(1024, 3)
ids = np.nonzero(w_all[:, 0] > 0)[0]; w_ = w_all[ids]
t0 = int(w_[:, 0].min()); st_, en_ = (w_[:, 0].astype(np.int64) - t0) / 100.0, (w_[:, 1].astype(np.int64) - t0) / 100.0
print("last launch: %d workgroups, entry %.2f .. %.2f us (median %.2f), end min %.2f median %.2f max %.2f us" % (len(w_), st_.min(), st_.max(), np.median(st_), en_.min(), np.median(en_), en_.max()))
cnt = (ctypes.c_uint * 4096)()
L.dtt_psroi_bwd_cnt_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert L.dtt_psroi_bwd_cnt_read(cnt, 4096)
cnt = np.array(cnt, dtype=np.int64).reshape(1024, 4)
print("list entries walked per workgroup: median %d max %d; entries taken serially (list overflow): total %d; chunks word (chunks | guessed << 8 | run << 16) of workgroup 0: %x"
      % (np.median(cnt[ids, 0]), cnt[ids, 0].max(), cnt[ids, 1].sum(), cnt[0, 3]))
life = en_ - st_
for k in np.argsort(-life)[:8]:
    hw = int(w_[k, 2] & np.uint64(0xffffffff)); xcc = int(w_[k, 2] >> np.uint64(32)) & 0xf
    print("   entries %d serial %d walks %d chunks %x" % tuple(cnt[ids[k]]), end="")
    print("   workgroup %4d: entry %.2f end %.2f us (%.2f us)  HW_ID %08x (cu %d sh %d se %d) XCC %d" % (ids[k], st_[k], en_[k], life[k], hw, (hw >> 8) & 0xf, (hw >> 12) & 1, (hw >> 13) & 7, xcc))
