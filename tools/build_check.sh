#!/bin/bash
# compile one source of the library with the kernel-resource remarks and print VGPRs / scratch per kernel
#   tools/build_check.sh mi32_blocked.hip [filter]
cd /root/repo/gpu_matrix_inversion_amd/csrc || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize \
  -Wall -Wno-unused-function -I../../include -I. $EXTRA -c "$1" -o /tmp/build_check.o -Rpass-analysis=kernel-resource-usage 2>/tmp/build_check.txt
grep -E "error|warning:" /tmp/build_check.txt | head -20
python3 /root/repo/tools/kernel_resources.py /tmp/build_check.txt "$2" | sort
