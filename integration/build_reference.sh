#!/bin/bash
# Builds the UNPATCHED reference (GATB-Core, /root/reference/gatb-core) with ITS OWN cmake into a scratch directory — nothing is written to the reference
# tree, no reference source enters this repository — and copies the tools the measurement and integration legs use into integration/_build/ref/
# (git-ignored; it travels to the GPU box with gpurun like the project's own built .so files):
#   dbgh5, dbginfo      bench.py's cpu_baseline leg, kind "reference": the reference's own multithreaded SortingCountAlgorithm timed on the GPU box's host cores
#   gatb-h5dump         tools/make_reference_run_vectors.py, integration/check_graphunitigs.sh
# and leaves lib/Release/libgatbcore.a + libhdf5.a in the build directory for integration/check_integration.sh --link.
#   integration/build_reference.sh [build dir, default /tmp/gatb_build]
# A build directory that already holds the artefacts is reused (the cmake build takes ~10 minutes on 8 cores).
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
REF=${GATB_REFERENCE:-/root/reference/gatb-core}
BUILD=${1:-/tmp/gatb_build}
test -d "$REF/src/gatb" || { echo "reference not found at $REF"; exit 3; }
if [ ! -f "$BUILD/lib/Release/libgatbcore.a" ] || [ ! -x "$BUILD/bin/Release/dbgh5" ] || [ ! -x "$BUILD/bin/Release/dbginfo" ] || [ ! -x "$BUILD/bin/Release/gatb-h5dump" ]; then
  mkdir -p "$BUILD"
  echo "[build_reference] configuring the reference in $BUILD"
  (cd "$BUILD" && cmake -DCMAKE_BUILD_TYPE=Release -DGATB_CORE_EXCLUDE_EXAMPLES=1 -DGATB_CORE_EXCLUDE_TESTS=1 "$REF" > cmake.log 2>&1) || { tail -20 "$BUILD/cmake.log"; exit 4; }
  echo "[build_reference] building (make -j$(nproc) dbgh5 dbginfo gatb-h5dump)"
  (cd "$BUILD" && make -j"$(nproc)" dbgh5 dbginfo gatb-h5dump > make.log 2>&1) || { tail -30 "$BUILD/make.log"; exit 5; }
fi
mkdir -p "$HERE/_build/ref"
for t in dbgh5 dbginfo gatb-h5dump; do cp -f "$BUILD/bin/Release/$t" "$HERE/_build/ref/$t"; done
echo "[build_reference] ok: $HERE/_build/ref/{dbgh5,dbginfo,gatb-h5dump} (from $BUILD)"
