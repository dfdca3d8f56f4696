#!/bin/bash
# Developer tool: libdtt_hip.so with -DDTT_WS_TRACE in the window-split correlation (per-workgroup phase stamps) into
# tools/_variants/wstrace.so; use with DTT_HIP_LIBRARY=tools/_variants/wstrace.so tools/ws_trace.py
set -e
cd "$(dirname "$0")/../pytorch-detect-to-track_amd/csrc"
make > /dev/null
mkdir -p ../../tools/_variants /tmp/dtt_ws_trace
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math \
  -I../../include -DDTT_WS_TRACE ${EXTRA_FLAGS} -c correlation_wsplit.hip -o /tmp/dtt_ws_trace/correlation_wsplit.o 2>/dev/null
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/_variants/wstrace.so $(ls build/*.o | grep -v correlation_wsplit) /tmp/dtt_ws_trace/correlation_wsplit.o -L/opt/rocm/lib -lhipblaslt
ls -la ../../tools/_variants/wstrace.so
