#!/bin/bash
# (GPU box) second probe call of round 6:  tests of the new paths, k = 63 timing, global-cursor scatter, the larger 8-rank dry run
set -u
OUT=gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_parity.py -q -x -k "packed_sink or raw_sink or top_word or counts_bit_exact or oversize or sliced or skewed_generator or fuzz or edge" 2>&1 | tail -5
python tools/fuzz_soak.py 100 200 2>&1 | tail -2
python tools/prof_step.py 100000000 8192 63 > $OUT/r06_k63_steps.txt 2>&1; grep -E "^iter" $OUT/r06_k63_steps.txt | cut -c1-700
python tools/prof_step.py 100000000 8192 63 1 > $OUT/r06_k63_steps_skewed.txt 2>&1; grep -E "^iter 2" $OUT/r06_k63_steps_skewed.txt | cut -c1-300
(cd tools/scatter_bench && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 scatter_bench.hip -o scatter_bench 2>/dev/null; ./scatter_bench 24 > ../../$OUT/r06_scatter_global_cursors.txt 2>&1); tail -8 $OUT/r06_scatter_global_cursors.txt
GKC_BENCH_BACKEND=gloo timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 8 --steps 3 --warmup 1 --reads 5000000 \
      > $OUT/r06_bench_8ranks_dryrun_5000000.json 2> $OUT/r06_bench_8ranks_dryrun_5000000.err
tail -c 300 $OUT/r06_bench_8ranks_dryrun_5000000.json; echo; grep -E "GkcError|out of memory" $OUT/r06_bench_8ranks_dryrun_5000000.err | head -3
