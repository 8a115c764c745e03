#!/bin/bash
# Regenerates the files under profiles/ on a GPU box (run through gpurun from the repo root):
#   kernel stats of the default bench, the two PMC passes (single lane, one step), then the bench lines themselves.
# Outputs land in gpurun_out/prof_round/ and are copied into profiles/ by hand afterwards.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/prof_round; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/kt -o r -- python $R/bench.py --no-cpu-baseline --no-host-landed --no-k63 --no-bloom-mphf --no-share-of-8 --no-two-pass > $O/kt.log 2>&1
GKC_STAGEB_LANES=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pf -o f -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-host-landed --no-k63 --no-bloom-mphf --no-share-of-8 --no-two-pass > $O/pf.log 2>&1
GKC_STAGEB_LANES=1 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pw -o w -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-host-landed --no-k63 --no-bloom-mphf --no-share-of-8 --no-two-pass > $O/pw.log 2>&1
cd $R
python tools/rocprof_summary.py $(find $O/kt -name "*.db" | head -n 1) > $O/kernel_stats.txt 2>&1
( echo "# PMC (separate rocprofv3 passes, GKC_STAGEB_LANES=1, --steps 1 --warmup 0: --pmc FETCH_SIZE | WRITE_SIZE on the default bench (1e8 reads); --kernel-trace only)"
  echo "# FETCH_SIZE / WRITE_SIZE in KiB; gfx950: FETCH_SIZE counts 1/2 of wide coalesced reads (MI355X_MICROARCH.md)"
  python tools/rocprof_summary.py $(find $O/pf -name "*.db" | head -n 1) | sed -n '/PMC counters/,$p'
  python tools/rocprof_summary.py $(find $O/pw -name "*.db" | head -n 1) | sed -n '/PMC counters/,$p' ) > $O/pmc.txt 2>&1
python tools/pmc_traffic.py $(find $O/pf -name "*.db" | head -n 1) $(find $O/pw -name "*.db" | head -n 1) "k=31, 100000000 synthetic 150 bp reads per GPU, single-pass count (no Bloom), m=10, 4096 partitions" $O/pmc_traffic.json > $O/pmc_traffic.log 2>&1
cp $O/pmc_traffic.json profiles/pmc_traffic.json
python bench.py > $O/bench_k31.json 2> $O/bench_k31.err

tail -c 400 $O/bench_k31.json; tail -n 5 $O/pmc_traffic.log; head -n 12 $O/kernel_stats.txt
