#!/bin/bash
# One GPU call of the development loop: (optional) GPU tests, the short bench (two lanes + its single-lane step), and a single-lane kernel trace.
#   gpurun -- 'bash tools/quick_round.sh <tag> [notest]'
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-q}; O=$R/gpurun_out/$TAG; rm -rf $O; mkdir -p $O
cd $R
if [ "$2" != "notest" ]; then python -m pytest tests -m gpu -x -q 2>&1 | tail -6; fi
python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-k63 --no-bloom-mphf --no-host-landed --no-share-of-8 --no-two-pass > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d = json.load(open("$O/bench.json"))
print("ms_per_step", d["ms_per_step"])
print("timed ", d["config"]["kernel_ms_per_step"])
print("single", d["roofline"]["single_lane"]["kernel_ms_per_step"])
PY
cd /tmp; export TMPDIR=/tmp
GKC_STAGEB_LANES=1 rocprofv3 --kernel-trace --stats -d $O/kt -o r -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-landed --no-k63 --no-bloom-mphf --no-share-of-8 --no-two-pass > $O/kt.log 2>&1
cd $R
python tools/rocprof_summary.py $(find $O/kt -name "*.db" | head -n 1) --seq 40 > $O/kernel_stats.txt 2>&1
rm -rf $O/kt
head -24 $O/kernel_stats.txt; tail -42 $O/kernel_stats.txt
