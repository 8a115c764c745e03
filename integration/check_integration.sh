#!/bin/bash
# Compile check of the reference-side binding (integration/gatb_device/*.hpp + gatb-core.device.patch) AGAINST THE REFERENCE.
#   integration/check_integration.sh [scratch dir] [--link [gatb build dir with lib/Release/libgatbcore.a]]
# 1. runs the reference's cmake CONFIGURE step in the scratch dir (generates gatb/system/api/config.hpp and the HDF5 configuration headers;
#    nothing is built, the reference tree is not written);
# 2. applies the patch to scratch copies of the seven files it touches (integration/make_patched_sources.py);
# 3. g++ -fsyntax-only of the reference's own template instantiation units that hold them — template/TemplateSpecialization1.cpp.in (ConfigurationAlgorithm),
#    2 (SortingCountAlgorithm + PartitionsCommand), 3 (BloomAlgorithm, DebloomAlgorithm, DebloomMinimizerAlgorithm), 4 (MPHFAlgorithm) — for spans 32
#    and 64 with -DGATB_WITH_DEVICE_COUNTING, plus explicit instantiations of PartitionsByDeviceCommand and of BloomDevice<LargeInt<1>>, <LargeInt<2>>;
# 4. --link: compiles those units for the four spans and links a patched dbgh5 (and integration/unitigs_check.cpp) against them + libgatbcore.a,
#    libhdf5.a and libgkc_hip.so (link only: running it needs a GPU) -> integration/_build/{dbgh5_device,unitigs_check}. The objects of the patched
#    units come first on the link line, so their definitions (explicit instantiations: weak symbols) are the ones kept. Without a built reference in
#    the given directory, integration/build_reference.sh builds it first (the reference's own cmake, ~10 minutes).
# This is a compile check in the build container, not an oracle: nothing it produces is used by the tests of the hot path.
set -e
HERE=$(cd "$(dirname "$0")" && pwd); REPO=$(dirname "$HERE")
REF=${GATB_REFERENCE:-/root/reference/gatb-core}
SCRATCH=${1:-/tmp/gkc_integration}; shift || true
LINK=0; LIBDIR=""
if [ "$1" == "--link" ]; then LINK=1; LIBDIR=${2:-/tmp/gatb_build}; fi
test -d "$REF/src/gatb" || { echo "reference not found at $REF"; exit 3; }
mkdir -p "$SCRATCH/cfg" "$SCRATCH/inc" "$SCRATCH/obj"
if [ ! -f "$SCRATCH/cfg/include/gatb/system/api/config.hpp" ]; then
  (cd "$SCRATCH/cfg" && cmake -DCMAKE_BUILD_TYPE=Release -DGATB_CORE_EXCLUDE_EXAMPLES=1 "$REF" > cmake.log 2>&1) || { tail -20 "$SCRATCH/cfg/cmake.log"; exit 4; }
fi
python3 "$HERE/make_patched_sources.py" "$REF" "$SCRATCH/inc" > /dev/null
# <hdf5/hdf5.h>: the reference's build copies the vendored HDF5 headers under include/Release/hdf5; a configure-only tree has them in two
# places (sources + the generated H5pubconf.h) -> a directory of symbolic links stands in for that copy step
mkdir -p "$SCRATCH/inc_hdf5/hdf5"
for h in "$REF"/thirdparty/hdf5/src/*.h "$SCRATCH"/cfg/thirdparty/hdf5/H5pubconf.h; do ln -sf "$h" "$SCRATCH/inc_hdf5/hdf5/$(basename "$h")"; done
INC="-I$SCRATCH/inc -I$SCRATCH/inc_hdf5 -I$HERE -I$REPO/include -I$SCRATCH/cfg/include -I$SCRATCH/cfg/include/Release -I$REF/src -I$REF/thirdparty"
FLAGS="-msse2 -msse4.2 -mpopcnt -std=c++11 -DNDEBUG -DINT128_FOUND -Wno-invalid-offsetof -Wno-unknown-pragmas -Wno-format -DGATB_WITH_DEVICE_COUNTING"
for K in 32 64 96 128; do
  { sed "s/\${KSIZE}/$K/g" "$REF/src/gatb/template/TemplateSpecialization2.cpp.in"
    echo "namespace gatb { namespace core { namespace kmer { namespace impl { template class PartitionsByDeviceCommand<$K>; } } } }"; } > "$SCRATCH/obj/ts2_$K.cpp.new"
  sed "s/\${KSIZE}/$K/g" "$REF/src/gatb/template/TemplateSpecialization1.cpp.in" > "$SCRATCH/obj/ts1_$K.cpp.new"
  sed "s/\${KSIZE}/$K/g" "$REF/src/gatb/template/TemplateSpecialization3.cpp.in" > "$SCRATCH/obj/ts3_$K.cpp.new"
  sed "s/\${KSIZE}/$K/g" "$REF/src/gatb/template/TemplateSpecialization4.cpp.in" > "$SCRATCH/obj/ts4_$K.cpp.new"
  for U in ts1 ts2 ts3 ts4; do cmp -s "$SCRATCH/obj/${U}_$K.cpp.new" "$SCRATCH/obj/${U}_$K.cpp" && rm "$SCRATCH/obj/${U}_$K.cpp.new" || mv "$SCRATCH/obj/${U}_$K.cpp.new" "$SCRATCH/obj/${U}_$K.cpp"; done
done
cat > "$SCRATCH/obj/bloom_device.cpp" <<EOT
#include <gatb/tools/collections/impl/Bloom.hpp>   /* the patched header: brings BloomDevice.hpp in */
#include <gatb/tools/math/LargeInt.hpp>
namespace gatb { namespace core { namespace tools { namespace collections { namespace impl {
template class BloomDevice<gatb::core::tools::math::LargeInt<1> >;
template class BloomDevice<gatb::core::tools::math::LargeInt<2> >;
} } } } }
EOT
echo "[check_integration] syntax: SortingCountAlgorithm + PartitionsByDeviceCommand, Bloom / Debloom, MPHF units (spans 32, 64), BloomDevice"
pids=""
SYN="ts1_32 ts1_64 ts2_32 ts2_64 ts3_32 ts3_64 ts4_32 ts4_64 bloom_device"
for f in $SYN; do g++ $FLAGS $INC -fsyntax-only "$SCRATCH/obj/$f.cpp" > "$SCRATCH/obj/$f.log" 2>&1 & pids="$pids $!"; done
rc=0; for p in $pids; do wait $p || rc=1; done
for f in $SYN; do grep -E "error" "$SCRATCH/obj/$f.log" | head -20 || true; done
[ $rc -eq 0 ] || { echo "[check_integration] SYNTAX CHECK FAILED"; exit 1; }
echo "[check_integration] syntax ok"
if [ $LINK -eq 1 ]; then
  # no built reference yet: integration/build_reference.sh builds it with the reference's own cmake (~10 minutes), so the chain is reproducible from this repository
  test -f "$LIBDIR/lib/Release/libgatbcore.a" || bash "$HERE/build_reference.sh" "$LIBDIR" || { echo "no libgatbcore.a under $LIBDIR and the reference build failed"; exit 5; }
  test -f "$REPO/gatb-core_amd/csrc/libgkc_hip.so" || { echo "libgkc_hip.so missing"; exit 6; }
  echo "[check_integration] compiling the patched instantiation units (4 units x 4 spans) and dbgh5"
  # an object is rebuilt when its unit, a patched source, a binding header or gkc.h is newer (the units take minutes each at -O2)
  newest=$(ls -t "$SCRATCH"/inc/gatb/kmer/impl/*.cpp "$SCRATCH"/inc/gatb/tools/collections/impl/*.hpp "$HERE"/gatb_device/*.hpp "$REPO/include/gkc.h" | head -1)
  OBJS=""; jobs_running=0; rc=0; pids=""
  for U in ts1 ts2 ts3 ts4; do for K in 32 64 96 128; do
    o="$SCRATCH/obj/${U}_$K.o"; OBJS="$OBJS $o"
    if [ ! -f "$o" ] || [ "$SCRATCH/obj/${U}_$K.cpp" -nt "$o" ] || [ "$newest" -nt "$o" ]; then
      g++ $FLAGS -O2 $INC -c "$SCRATCH/obj/${U}_$K.cpp" -o "$o" > "$SCRATCH/obj/${U}_$K.clog" 2>&1 & pids="$pids $!"
      jobs_running=$((jobs_running+1))
      if [ $jobs_running -ge ${GKC_INTEGRATION_JOBS:-$(nproc)} ]; then for p in $pids; do wait $p || rc=1; done; pids=""; jobs_running=0; fi
    fi
  done; done
  if [ ! -f "$SCRATCH/obj/dbgh5.o" ]; then g++ $FLAGS -O1 $INC -c "$REF/tools/dbgh5.cpp" -o "$SCRATCH/obj/dbgh5.o" > "$SCRATCH/obj/dbgh5.clog" 2>&1 & pids="$pids $!"; fi
  g++ $FLAGS -O1 $INC -c "$HERE/unitigs_check.cpp" -o "$SCRATCH/obj/unitigs_check.o" > "$SCRATCH/obj/unitigs_check.clog" 2>&1 & pids="$pids $!"
  for p in $pids; do wait $p || rc=1; done
  [ $rc -eq 0 ] || { grep -h error "$SCRATCH"/obj/*.clog | head; echo "[check_integration] COMPILE FAILED"; exit 1; }
  LIBS="$LIBDIR/lib/Release/libgatbcore.a $LIBDIR/lib/Release/libhdf5.a -L$REPO/gatb-core_amd/csrc -lgkc_hip -Wl,-rpath,$REPO/gatb-core_amd/csrc -Wl,-rpath,/opt/rocm/lib -ldl -lpthread -lz -lm"
  g++ -o "$SCRATCH/dbgh5_device" "$SCRATCH/obj/dbgh5.o" $OBJS $LIBS > "$SCRATCH/obj/link.log" 2>&1 || { head -30 "$SCRATCH/obj/link.log"; echo "[check_integration] LINK FAILED"; exit 1; }
  g++ -o "$SCRATCH/unitigs_check" "$SCRATCH/obj/unitigs_check.o" $OBJS $LIBS > "$SCRATCH/obj/link_unitigs.log" 2>&1 || { head -30 "$SCRATCH/obj/link_unitigs.log"; echo "[check_integration] LINK (unitigs_check) FAILED"; exit 1; }
  # ... and the same driver against the UNPATCHED library only: the reference's GraphUnitigs as the consumer of an .h5 (integration/_build/ref/, beside the reference's tools)
  g++ -o "$SCRATCH/unitigs_check_ref" "$SCRATCH/obj/unitigs_check.o" "$LIBDIR/lib/Release/libgatbcore.a" "$LIBDIR/lib/Release/libhdf5.a" -ldl -lpthread -lz -lm > "$SCRATCH/obj/link_unitigs_ref.log" 2>&1 \
      || { head -30 "$SCRATCH/obj/link_unitigs_ref.log"; echo "[check_integration] LINK (unitigs_check, reference only) FAILED"; exit 1; }
  mkdir -p "$HERE/_build/ref" && cp -f "$SCRATCH/unitigs_check_ref" "$HERE/_build/ref/unitigs_check"
  nm -C "$SCRATCH/dbgh5_device" | grep -c "PartitionsByDeviceCommand" | sed 's/^/[check_integration] PartitionsByDeviceCommand symbols in the patched dbgh5: /'
  nm -D "$SCRATCH/dbgh5_device" | grep -E " U gkc_" | sed 's/^/[check_integration] imports /'
  mkdir -p "$HERE/_build" && cp -f "$SCRATCH/dbgh5_device" "$SCRATCH/unitigs_check" "$HERE/_build/"      # git-ignored; travels to the GPU box with gpurun (tools/run_patched_dbgh5.py, tests/test_gpu_dropin.py)
  echo "[check_integration] link ok: $SCRATCH/dbgh5_device, unitigs_check (copied to integration/_build/)"
fi
