#!/bin/bash
# Developer tool: the whole library with -DDTT_WG_TRACE (per-workgroup placement / phase stamps in proposal_select_sort) into tools/_variants/wgtrace.so; use with DTT_HIP_LIBRARY=tools/_variants/wgtrace.so tools/wg_trace.py
set -e
cd "$(dirname "$0")/../pytorch-detect-to-track_amd/csrc"
B=/tmp/dtt_build_trace; mkdir -p $B ../../tools/_variants
for f in $(grep "^SRCS" Makefile | sed 's/SRCS := //'); do
  if [ ! -f $B/${f%.hip}.o ] || [ $f -nt $B/${f%.hip}.o ] || [ "$f" = "proposal.hip" ]; then
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math \
      -I../../include -DDTT_WG_TRACE ${EXTRA_FLAGS} -c $f -o $B/${f%.hip}.o 2>/dev/null &
  fi
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/_variants/wgtrace.so $B/*.o -L/opt/rocm/lib -lhipblaslt
ls -la ../../tools/_variants/wgtrace.so
