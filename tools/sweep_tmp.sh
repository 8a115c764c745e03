python -m pytest tests -m gpu -x -q 2>&1 | tail -5
bash tools/profile_round.sh > gpurun_out/prof_round.log 2>&1; tail -15 gpurun_out/prof_round.log
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/prof_k63; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/kt -o r -- python $R/bench.py --k 63 --partitions 8192 --steps 2 --warmup 1 --no-cpu-baseline --no-host-landed > $O/kt.log 2>&1
cd $R
python tools/rocprof_summary.py $(find $O/kt -name "*.db" | head -n 1) > $O/kernel_stats_k63.txt 2>&1
rm -rf $O/kt; head -14 $O/kernel_stats_k63.txt
