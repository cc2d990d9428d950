#!/bin/bash
# Timing experiments on the 256 x 256 weight-gradient kernel: builds libmvuld_hip variants with -DTN256_X=n (see gemm_tn256.hip; the
# variants compute WRONG results, they only tell where the time goes) into build_variants/.  Run tools/gemm_shapes.py --only tn with
# MVULD_HIP_LIB pointing at each.
set -e
cd "$(dirname "$0")/../mvuld_amd/csrc"
mkdir -p ../../build_variants
for x in 1 2 4; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -Wno-unused-result -DTN256_X=$x -c gemm_tn256.hip -o /tmp/tn256_x$x.o
  objs=$(ls build/*.o | grep -v gemm_tn256.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build_variants/libmvuld_tn$x.so $objs /tmp/tn256_x$x.o
done
