#!/bin/bash
# Build a variant of the library next to the product one: tools/build_variant.sh libiqlhip_stamps.so -DIQL_STAMPS
# (the flags of __graft_entry__.build() plus the extra ones; used for A/B runs and the s_memtime stamp build)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-unused-value -mllvm -amdgpu-kernarg-preload-count=14 \
  -I "$ROOT/include" "$@" -o "$ROOT/jsrl-corl_amd/$OUT" "$ROOT/jsrl-corl_amd/csrc/iqlhip.hip"
