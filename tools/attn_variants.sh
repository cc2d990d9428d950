#!/bin/bash
# Timing experiments on the attention kernels: builds libmvuld_hip variants with -DAM_X=n (see attention_mfma.hip) into build_variants/.
set -e
cd "$(dirname "$0")/../mvuld_amd/csrc"
for x in ${AM_VARIANTS:-1 2 3}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -Wno-unused-result -DAM_X=$x -c attention_mfma.hip -o /tmp/attn_x$x.o
  objs=$(ls build/*.o | grep -v attention_mfma.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build_variants/libmvuld_x$x.so $objs /tmp/attn_x$x.o
done
