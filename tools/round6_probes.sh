#!/bin/bash
# (GPU box) Round-6 probes in ONE call:  bash tools/round6_probes.sh   -> gpurun_out/r06_*
#   1. bench.py dry runs at 8 ranks on the one GPU (gloo-backed transport): 10^6 reads per rank, and 5e6 per rank (4e7 in all: 8 contexts share the one GPU's memory) with the sink-mode trial
#   2. frequency-order minimizers: L2 (TCC) hit / miss / request counters of k_scan_tile in both modes
#   3. the headline's lead-in: first Stage-B batch of a lane 1/4 (round 5), 1/8, 1/16 of the others
set -u
OUT=gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
LEAN="--no-k63 --no-two-pass --no-share-of-8 --no-cpu-baseline --no-bloom-mphf --no-freq-order --no-skewed"
# ---- 1
for R in 1000000 5000000; do
  GKC_BENCH_BACKEND=gloo timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 8 --steps 3 --warmup 1 --reads $R \
      > $OUT/r06_bench_8ranks_dryrun_${R}.json 2> $OUT/r06_bench_8ranks_dryrun_${R}.err
  tail -c 300 $OUT/r06_bench_8ranks_dryrun_${R}.json; echo; grep -E "communicator|sink" $OUT/r06_bench_8ranks_dryrun_${R}.err | head -5
done
# ---- 2
rocprofv3 -L 2>/dev/null | grep -oE "TCC_[A-Z_]*(HIT|MISS|REQ)[A-Za-z_]*|TCP_TCC_READ_REQ[a-z_]*" | sort -u | head -40 > $OUT/r06_tcc_counters_available.txt
for MODE in 0 1; do
  for C in "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
    TAG=$(echo $C | tr ' ' '+')
    timeout 900 rocprofv3 --pmc $C --kernel-trace -d $OUT/r06_l2_${MODE}_$TAG -o r -- python tools/prof_step.py 100000000 4096 31 0 $MODE > $OUT/r06_l2_${MODE}_$TAG.log 2>&1
    DB=$(find $OUT/r06_l2_${MODE}_$TAG -name "*.db" | head -1)
    { echo "## freq_mode=$MODE  --pmc $C"; [ -n "$DB" ] && python tools/rocprof_summary.py $DB | grep -E "k_scan_tile|k_emit_desc|^# PMC|^kernel" ; grep -E "^iter 2" $OUT/r06_l2_${MODE}_$TAG.log | cut -c1-400; } >> $OUT/r06_freq_order_l2_raw.txt
    rm -rf $OUT/r06_l2_${MODE}_$TAG
  done
done
# ---- 3
for DIV in 4 8 16 32; do
  GKC_SINK_FIRST_DIV=$DIV python bench.py --steps 8 --warmup 2 $LEAN > $OUT/r06_leadin_div$DIV.json 2> /dev/null
  python - <<PY
import json
d = json.load(open("$OUT/r06_leadin_div$DIV.json"))
print("GKC_SINK_FIRST_DIV=$DIV  ms_per_step", round(d["ms_per_step"], 1), "median", d.get("ms_per_step_median"), "device-resident", round(d.get("ms_per_step_device_resident", 0), 1), "wire GB", d.get("landed", {}).get("bytes_over_the_link"), "verified", d.get("all_blocks_verified"))
PY
done
