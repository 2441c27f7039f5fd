#!/bin/bash
# bash profiles/refresh_r04.sh : the round's bench lines, microbenchmarks and timelines.  Output under gpurun_out/r04/
# (copied into profiles/r04/ afterwards).
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r04
mkdir -p $O
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -5 $O/bench_default.err; exit 1; }
python3 bench.py --mode train --no-cpu-baseline --no-extra-legs --steps 200 --warmup 20 > $O/bench_train.json 2> $O/bench_train.err || { tail -5 $O/bench_train.err; exit 1; }
python3 bench.py --workload c4 > $O/bench_c4_4096x256.json 2> $O/bench_c4.err || { tail -5 $O/bench_c4.err; exit 1; }
python3 bench.py --workload c4 --c4-point 512,32 --steps 256 --warmup 32 > $O/bench_c4_512x32.json 2> $O/bench_c4s.err || { tail -5 $O/bench_c4s.err; exit 1; }
echo "bench lines done"
python3 profiles/step_timeline.py c2 > $O/timeline_r04_c2.txt 2> $O/timeline.err || { tail -5 $O/timeline.err; exit 1; }
python3 profiles/step_timeline.py c2adam > $O/timeline_r04_c2adam.txt 2>> $O/timeline.err || { tail -5 $O/timeline.err; exit 1; }
python3 profiles/bench_tail.py > $O/bench_tail.log 2>&1 || { tail -5 $O/bench_tail.log; exit 1; }
python3 profiles/microbench.py > $O/microbench_final.log 2>&1 || { tail -5 $O/microbench_final.log; exit 1; }
python3 profiles/sweep_l1_bwd.py > $O/sweep_l1_bwd.log 2>&1 || { tail -5 $O/sweep_l1_bwd.log; exit 1; }
python3 profiles/bench_index.py > $O/bench_index.log 2>&1 || { tail -5 $O/bench_index.log; exit 1; }
python3 profiles/bench_topk.py > $O/bench_topk.log 2>&1 || { tail -5 $O/bench_topk.log; exit 1; }
echo "all done"
