#!/bin/bash
# Timing experiments on the persistent NT GEMM: builds libmvuld_hip variants with -DP256_X=n (see gemm_p256.hip) into build_variants/.
set -e
cd "$(dirname "$0")/../mvuld_amd/csrc"
mkdir -p ../../build_variants
for x in 1 2 3; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -Wno-unused-result -DP256_X=$x -c gemm_p256.hip -o /tmp/p256_x$x.o
  objs=$(ls build/*.o | grep -v gemm_p256.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build_variants/libmvuld_p$x.so $objs /tmp/p256_x$x.o
done
