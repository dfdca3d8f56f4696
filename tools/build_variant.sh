#!/bin/bash
# Developer tool: build a variant of libdtt_hip.so with extra -D flags into tools/_variants/<name>.so
#   tools/build_variant.sh stamp "-DDTT_CORR_STAMP"            (correlation only, default)
#   SRCS="common.hip psroi.hip" tools/build_variant.sh ct1 "-DDTT_PSROI_CT=1"
set -e
cd "$(dirname "$0")/../pytorch-detect-to-track_amd/csrc"
mkdir -p ../../tools/_variants
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math \
  -I../../include $2 -shared -o ../../tools/_variants/$1.so ${SRCS:-common.hip correlation.hip}
