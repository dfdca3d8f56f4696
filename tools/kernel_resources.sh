#!/bin/bash
# Registers / scratch / static LDS / occupancy of every kernel in libdtt_hip.so (hipcc -Rpass-analysis=kernel-resource-usage,
# the Makefile's flags): tools/kernel_resources.sh > profiles/rNN_kernel_resources.txt
cd "$(dirname "$0")/../pytorch-detect-to-track_amd/csrc"
echo "# hipcc --offload-arch=gfx950 -O3 -Rpass-analysis=kernel-resource-usage over csrc/*.hip (same flags as the Makefile): registers, scratch (spills),"
echo "# static LDS and occupancy (waves per SIMD) of every kernel in libdtt_hip.so.  LDS_B is the STATIC allocation only: the correlation, head GEMM,"
echo "# PSRoI, NMS, proposal, RoI Align and tube kernels size their LDS at launch (DESIGN.md section 3)."
printf "%-100s %6s %6s %9s %9s %5s\n" kernel VGPRs AGPRs scratch_B LDS_B occ
for f in $(grep "^SRCS" Makefile | sed 's/SRCS := //'); do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math \
    -I../../include -Rpass-analysis=kernel-resource-usage -c $f -o /dev/null 2>&1 | grep "remark:" | sed 's/ \[-Rpass-analysis=kernel-resource-usage\]//' | awk '
    /Function Name:/ {name=$NF}
    / VGPRs:/ {v=$NF}
    / AGPRs:/ {a=$NF}
    /ScratchSize/ {s=$NF}
    /Occupancy/ {o=$NF}
    /LDS Size/ {l=$NF; printf "%-100s %6s %6s %9s %9s %5s\n", substr(name,1,100), v, a, s, l, o}'
done
