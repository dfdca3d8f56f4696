#!/usr/bin/env bash
# oracle/_ref: the reference's OWN kernels, built from the sources where they lie under /root/reference.
#
# The six operator kernels of the hot path (correlation, PSRoI pooling, NMS, RoI align / pool / crop) are
# self-contained CUDA translation units: each includes only the C library and its own header, and exposes plain
# extern "C" launchers that take raw device pointers (the functions the reference's TH/cffi shims call).  ROCm ships
# AMD's CUDA->HIP source translator (hipify-perl, /opt/rocm/bin), so those translation units can be compiled for
# gfx950 without writing a single stand-in header or line of code:
#     hipify-perl  <reference>.cu / .h   ->  temporary directory (deleted afterwards; nothing is kept or committed)
#     sed: the kernel-launch chevrons "<< <" / ">> >" that nvcc tolerates are closed up (token spacing only)
#     hipcc --offload-arch=gfx950        ->  oracle/_ref/libdtt_ref_kernels.so        (-ffp-contract=off: the declared
#                                            semantics of DESIGN.md section 2, what the oracle restates bit for bit)
#                                            oracle/_ref/libdtt_ref_kernels_fma.so    (hipcc's default contraction, the
#                                            analogue of nvcc's -fmad=true; used to show where contraction matters)
# The TH/THC shims (*_cuda.c) and the CPU variants (roi_pooling.c, roi_crop.c) need TH.h / THC.h and are NOT built.
# The library is test infrastructure: tests/test_gpu_ref_kernels.py runs these kernels on the MI355X beside the oracle
# and beside libdtt_hip.so.  oracle/_ref/ is git-ignored (only the .so travels to the GPU box); the recipe does nothing
# when /root/reference or hipify-perl is absent.
set -euo pipefail
REF="${DTT_REFERENCE_ROOT:-/root/reference}/lib/model"
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/_ref"
HIPIFY="${HIPIFY:-/opt/rocm/bin/hipify-perl}"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
if [ ! -d "$REF" ] || [ ! -x "$HIPIFY" ]; then
  echo "[oracle/_ref] reference sources or hipify-perl not present: nothing built"; exit 0
fi
UNITS="correlation/src/correlation_cuda_kernel psroi_pooling/src/psroi_pooling_kernel nms/src/nms_cuda_kernel
       roi_align/src/roi_align_kernel roi_pooling/src/roi_pooling_kernel roi_crop/src/roi_crop_cuda_kernel"
mkdir -p "$OUT"
TMP="$(mktemp -d "$OUT/tmp.XXXXXX")"
trap 'rm -rf "$TMP"' EXIT
for u in $UNITS; do
  b="$(basename "$u")"
  "$HIPIFY" "$REF/$u.h" > "$TMP/$b.h" 2>/dev/null
  "$HIPIFY" "$REF/$u.cu" 2>/dev/null | sed -E 's/<<[[:space:]]+</<<</g; s/>>[[:space:]]+>/>>>/g' > "$TMP/$b.hip"
done
build() {  # $1 = output name, rest = extra flags
  local so="$1"; shift
  local objs=""
  for u in $UNITS; do
    b="$(basename "$u")"
    # <cstring>: nvcc's implicit includes provide memset for the NMS host sweep
    # -fhip-fp32-correctly-rounded-divide-sqrt: IEEE division, nvcc's default (-prec-div=true)
    "$HIPCC" --offload-arch=gfx950 -O2 -fPIC -w -I"$TMP" -include cstring -fhip-fp32-correctly-rounded-divide-sqrt \
      "$@" -c "$TMP/$b.hip" -o "$TMP/$b.o"
    objs="$objs $TMP/$b.o"
  done
  "$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$OUT/$so" $objs
  echo "[oracle/_ref] built $OUT/$so"
}
build libdtt_ref_kernels.so -ffp-contract=off
build libdtt_ref_kernels_fma.so -ffp-contract=fast
