#!/bin/bash
# Builds an experiment variant of libgkc_hip.so with extra -D flags (timing experiments: a phase disabled, another tile size ...).
#   tools/build_variant.sh NAME -DGKC_EXP_NOSORT=1 ...   ->  gatb-core_amd/csrc/variants/libgkc_hip_NAME.so
# Run with GKC_LIB=<that file> python bench.py ...   Variants are never shipped or tested; results of a variant may be wrong by design.
set -e
cd "$(dirname "$0")/../gatb-core_amd/csrc"
name=$1; shift
rm -rf variants/$name; mkdir -p variants/$name
for f in gkc_api gkc_scan gkc_count gkc_sink gkc_bloom gkc_fastx gkc_mphf gkc_dist; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result "$@" -c $f.hip -o variants/$name/$f.o &
done
wait; for f in gkc_api gkc_scan gkc_count gkc_sink gkc_bloom gkc_fastx gkc_mphf gkc_dist; do test -f variants/$name/$f.o || { echo "compile of $f failed"; exit 1; }; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libgkc_hip_$name.so variants/$name/*.o -ldl
echo built variants/libgkc_hip_$name.so
