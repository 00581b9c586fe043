#!/bin/bash
# kernel resource usage of one HIP translation unit (cross-compiles on CPU): tools/kres.sh pairwise.hip [filter] [extra flags]
f=$1; filt=${2:-.}; shift; shift
cd /tmp && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I/root/repo/include -I/root/repo/tidypopgen_amd/csrc "$@" \
  -Rpass-analysis=kernel-resource-usage --cuda-device-only -c /root/repo/tidypopgen_amd/csrc/$f -o /tmp/kres.o 2>&1 |
  grep -E "Function Name|VGPRs:|AGPRs:|Spill|ScratchSize|LDS Size" | sed -E 's/.*remark: +//; s/ \[-Rpass.*//' |
  awk '/Function Name/{if(n)print n; n=$0; next}{n=n" | "$0}END{print n}' | sed 's/Function Name: //' | c++filt | grep -E "$filt" | cut -c1-260
