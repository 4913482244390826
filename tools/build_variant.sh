#!/bin/bash
# tools/build_variant.sh <tag> [-DFLAG=...]...  ->  variants/lib_<tag>.so : a build of the HIP library with extra compile-time
# switches, from a snapshot of the sources (so that the tree can change while hipcc runs).  For tools/gpu_ab.sh.
set -e
tag=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
work=$(mktemp -d /tmp/ark355_variant_XXXX)
mkdir -p "$work/snark_amd"; cp -r "$root/snark_amd/csrc" "$work/snark_amd/csrc"; cp -r "$root/include" "$work/include"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -DNDEBUG -Wno-unused-result"
pids=()
for s in capi ark355_bls ark355_bn; do
  hipcc $FLAGS "$@" -I "$work/snark_amd/csrc" -c "$work/snark_amd/csrc/$s.hip" -o "$work/$s.o" & pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
mkdir -p "$root/variants"
hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-Bsymbolic -o "$root/variants/lib_$tag.so" "$work"/*.o -L/opt/rocm/lib -lrccl
rm -rf "$work"
echo "$root/variants/lib_$tag.so"
