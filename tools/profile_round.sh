#!/bin/bash
# (GPU box) The round's committed measurements from ONE call, so that the kernel table, the PMC traffic and the bench line quote the same box and library:
#   tools/profile_round.sh <tag, e.g. r05>      -> gpurun_out/<tag>_*  (copied into profiles/ afterwards)
set -u
TAG=${1:-r05}; OUT=gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
# VRAM touched once (a fresh box pays ~26 ms per GB the first time a region is allocated: tools/alloc_probe)
[ -x tools/alloc_probe/alloc_probe ] && tools/alloc_probe/alloc_probe malloc 16 16 > $OUT/${TAG}_vram_touch.txt 2>&1
# 1. the driver's line
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/${TAG}_bench_1gpu_1e8reads.json 2> $OUT/${TAG}_bench.err
# 2. kernel trace of the headline blocks (two lanes, host-landed region + device-resident region), k = 31
LEAN="--no-k63 --no-two-pass --no-share-of-8 --no-cpu-baseline --no-bloom-mphf --no-freq-order --no-skewed"
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_kt -o r -- python bench.py --steps 5 --warmup 2 $LEAN > $OUT/${TAG}_kt_line.json 2> /dev/null
python tools/rocprof_summary.py $(find $OUT/${TAG}_kt -name "*.db" | head -1) > $OUT/${TAG}_kernel_stats_1e8reads.txt
# 3. k = 63, two lanes (tools/prof_step.py: 3 steps, the first one cold)
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_kt63 -o r -- python tools/prof_step.py 100000000 8192 63 > $OUT/${TAG}_k63_steps.txt 2> /dev/null
python tools/rocprof_summary.py $(find $OUT/${TAG}_kt63 -name "*.db" | head -1) > $OUT/${TAG}_kernel_stats_k63_1e8reads.txt
# 4. PMC: separate passes, one Stage-B lane, records left in HBM; the command runs 2 steps (1 untimed + 1 timed)
for C in FETCH_SIZE WRITE_SIZE; do
  GKC_STAGEB_LANES=1 GKC_BENCH_VALUE=device rocprofv3 --pmc $C --kernel-trace -d $OUT/${TAG}_pmc_$C -o r -- python bench.py --steps 1 --warmup 0 $LEAN --no-host-landed > $OUT/${TAG}_pmc_${C}_line.json 2> /dev/null
done
WL=$(python -c "import json;print(json.load(open('$OUT/${TAG}_pmc_FETCH_SIZE_line.json'))['config']['workload'])")
python tools/pmc_traffic.py $(find $OUT/${TAG}_pmc_FETCH_SIZE -name "*.db" | head -1) $(find $OUT/${TAG}_pmc_WRITE_SIZE -name "*.db" | head -1) "$WL" $OUT/${TAG}_pmc_traffic.json 2 > /dev/null
{ echo "# PMC (separate rocprofv3 passes, GKC_STAGEB_LANES=1, records left in HBM, 2 steps per pass: --pmc FETCH_SIZE | WRITE_SIZE with --kernel-trace only); KiB"; 
  python tools/rocprof_summary.py $(find $OUT/${TAG}_pmc_FETCH_SIZE -name "*.db" | head -1) | sed -n '/PMC counters/,$p'; python tools/rocprof_summary.py $(find $OUT/${TAG}_pmc_WRITE_SIZE -name "*.db" | head -1) | sed -n '/PMC counters/,$p'; } > $OUT/${TAG}_pmc.txt
rm -rf $OUT/${TAG}_kt $OUT/${TAG}_kt63 $OUT/${TAG}_pmc_FETCH_SIZE $OUT/${TAG}_pmc_WRITE_SIZE
python - <<PY
import json
d=json.load(open("$OUT/${TAG}_bench_1gpu_1e8reads.json"))
print({k:d.get(k) for k in ("value","ms_per_step","ms_per_step_median","value_device_resident","ms_per_step_device_resident","all_blocks_verified")})
print("roofline", {k:d["roofline"].get(k) for k in ("kernel","frac","launch_ms","traffic")}, d["roofline"].get("device_resident_region"))
for b in ("freq_order","skewed","k63","share_of_8","bloom_mphf","two_pass_overlap"):
    x=d["config"].get(b,{}); print(b, {k:x.get(k) for k in ("ms_per_step","value","verified","bloom_contains8_ms","value_host_landed")})
print("cpu", {k:d.get("cpu_baseline",{}).get(k) for k in ("value","kind","cores","dsk_time_s")})
PY
