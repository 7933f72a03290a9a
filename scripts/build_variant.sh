#!/bin/bash
# Builds an A/B variant of the product library with extra hipcc flags into nova_amd/libnova_mi355x_<name>.so
# (selected at run time with NMX_SO=...).  usage: scripts/build_variant.sh nolat -DNMX_LAT_TAIL=0
set -e
name=$1; shift
cd "$(dirname "$0")/.."
mkdir -p build/$name
pids=()
for u in nova_amd/csrc/*.hip; do
  o=build/$name/$(basename ${u%.hip}).o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c $u -o $o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o nova_amd/libnova_mi355x_$name.so build/$name/*.o
echo built nova_amd/libnova_mi355x_$name.so
