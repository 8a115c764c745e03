#!/bin/bash
# GraphUnitigs / bcalm2 consumption check. Build container only (needs the reference sources and a built libgatbcore.a).
#   integration/check_graphunitigs.sh <ours.h5 written by gkc_dsk on the GPU box> <the FASTA it was counted from> [gatb build dir] [k] [abundance-min]
# 1. builds integration/unitigs_check.cpp against the reference library;
# 2. lets the reference's own dbgh5 count the same FASTA in the mode GraphUnitigs forces (-minimizer-type 1 -repartition-type 1) -> ref.h5;
# 3. runs GraphUnitigs on ours.h5 and on ref.h5 and compares the unitigs as sets of canonical sequences;
# 4. also lets the reference's dbgh5 finish the legacy graph on ours.h5 (bloom, debloom, branching, mphf: state 7 -> 127).
set -e
HERE=$(cd "$(dirname "$0")" && pwd); REPO=$(dirname "$HERE")
OURS=$(readlink -f "$1"); FASTA=$(readlink -f "$2"); BUILD=${3:-/tmp/gatb_build}; K=${4:-21}; AMIN=${5:-2}
REF=${GATB_REFERENCE:-/root/reference/gatb-core}
W=${GKC_UNITIGS_SCRATCH:-/tmp/gkc_unitigs}; rm -rf "$W"; mkdir -p "$W"; cd "$W"
CFG=/tmp/gkc_integration
test -f "$CFG/cfg/include/gatb/system/api/config.hpp" || bash "$HERE/check_integration.sh" "$CFG" > /dev/null
INC="-I$CFG/inc_hdf5 -I$CFG/cfg/include -I$REF/src -I$REF/thirdparty"
g++ -std=c++11 -O1 -DNDEBUG -DINT128_FOUND -msse4.2 -mpopcnt -Wno-invalid-offsetof $INC "$HERE/unitigs_check.cpp" -o unitigs_check \
    "$BUILD/lib/Release/libgatbcore.a" "$BUILD/lib/Release/libhdf5.a" -ldl -lpthread -lz -lm
echo "[graphunitigs] driver built"
"$BUILD/bin/Release/dbgh5" -in "$FASTA" -kmer-size $K -abundance-min $AMIN -out ref -out-tmp . -nb-cores 1 -max-memory 1 -verbose 0 \
    -bloom none -debloom none -branching-nodes none -no-mphf -minimizer-type 1 -repartition-type 1 > ref_count.log 2>&1
cp "$OURS" ours.h5; cp ours.h5 ours_legacy.h5
canon() { python3 - "$1" <<'EOP'
import sys
comp = str.maketrans("ACGT", "TGCA")
seqs = []
for line in open(sys.argv[1]):
    if not line.startswith(">"):
        s = line.strip(); r = s.translate(comp)[::-1]; seqs.append(min(s, r))
seqs.sort()
import hashlib
print(len(seqs), sum(len(s) for s in seqs), hashlib.sha256("\n".join(seqs).encode()).hexdigest())
EOP
}
./unitigs_check ours.h5 ours_u 1 > ours_unitigs.log 2>&1 || { tail -5 ours_unitigs.log; echo "[graphunitigs] GraphUnitigs FAILED on ours.h5"; exit 1; }
./unitigs_check ref.h5  ref_u  1 > ref_unitigs.log 2>&1
A=$(canon ours_u.unitigs.fa); B=$(canon ref_u.unitigs.fa)
echo "[graphunitigs] unitigs from ours.h5 : $A   (count, total length, sha256 of the sorted canonical sequences)"
echo "[graphunitigs] unitigs from ref.h5  : $B"
[ "$A" == "$B" ] && echo "[graphunitigs] IDENTICAL unitig sets" || { echo "[graphunitigs] unitig sets DIFFER"; exit 1; }
"$BUILD/bin/Release/dbgh5" -in ours_legacy.h5 -out-tmp . -nb-cores 1 -verbose 0 > legacy.log 2>&1 || { tail -5 legacy.log; echo "[graphunitigs] dbgh5 -in ours.h5 FAILED"; exit 1; }
echo "[graphunitigs] dbgh5 -in ours.h5 completed the legacy graph: state $("$BUILD/bin/Release/gatb-h5dump" -a /state ours_legacy.h5 | grep '(0)' | tr -d ' ')"
