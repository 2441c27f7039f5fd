#!/bin/bash
# bash profiles/refresh_r03.sh : re-measure what changed late in round r03 (shared-negative L1 backward, ranks in the
# scoring epilogue) - bench lines, step traces, microbenchmarks, counters.  Output under gpurun_out/r03/.
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r03
mkdir -p $O
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -5 $O/bench_default.err; exit 1; }
python3 bench.py --workload c4 > $O/bench_c4_4096x256.json 2> $O/bench_c4.err || { tail -5 $O/bench_c4.err; exit 1; }
python3 bench.py --workload c4 --c4-point 512,32 --steps 256 --warmup 32 > $O/bench_c4_512x32.json 2> $O/bench_c4s.err || { tail -5 $O/bench_c4s.err; exit 1; }
echo "bench lines done"
bash profiles/run_step_traces.sh r03 "c4s c4g c4 c4n2" || exit 1
echo "traces done"
python3 profiles/microbench.py > $O/microbench_final.log 2>&1 || { tail -5 $O/microbench_final.log; exit 1; }
python3 profiles/sweep_l1_bwd.py > $O/sweep_l1_bwd.log 2>&1 || { tail -5 $O/sweep_l1_bwd.log; exit 1; }
python3 profiles/bench_topk.py > $O/bench_topk.log 2>&1 || { tail -5 $O/bench_topk.log; exit 1; }
python3 profiles/stress_shared_distance.py > $O/stress_shared_distance.log 2>&1 || { tail -5 $O/stress_shared_distance.log; exit 1; }
echo "microbenchmarks done"
bash profiles/pmc_l1_bwd.sh > $O/pmc_l1_kernels.txt 2>&1 || { tail -5 $O/pmc_l1_kernels.txt; exit 1; }
echo "all done"
