#!/bin/bash
# Profiling recipe for one round (run on the GPU box through gpurun):
#   bash profiles/run_profiles.sh r01
# Writes raw rocprofv3 output under gpurun_out/prof_<tag>/; the summaries that
# are judged are copied from there into profiles/ (tracked).
set -o pipefail
TAG=${1:-r01}
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
# the headline leg only: the extra legs of the bench line (roofline_hbm, c4, train_step) have their own runs below
BENCH="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs"
# 1. kernel trace + stats, score mode (the bench default) and train mode
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/score -- $BENCH > $OUT/score.json 2> $OUT/score.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train -- $BENCH --mode train > $OUT/train.json 2> $OUT/train.err || exit 1
# 2. HBM traffic counters, separate passes (TCC slots: FETCH_SIZE and WRITE_SIZE do not fit together)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT/pmc_write.json 2> $OUT/pmc_write.err || exit 1
# 3. out-of-cache variant of the same kernel (8 GB shard: every row comes from HBM)
$BENCH --entities-per-shard 4000000 > $OUT/score_hbm.json 2> $OUT/score_hbm.err || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_hbm -- $BENCH --entities-per-shard 4000000 > $OUT/pmc_fetch_hbm.json 2> $OUT/pmc_fetch_hbm.err || exit 1
find $OUT -name "*.csv" | head -50
# 4. one traced training step per workload (kernel by kernel, in start order): C2 (SGD, AdamW), C4 (S=512 eager /
#    hipGraph / step plan, two shards in lock-step eager / hipGraph, S=4096), the headline scoring step and the ScoreMoving form of the C2 training step
bash profiles/run_step_traces.sh $TAG "c2score c2 c2adam c2sm c2em2 c4s c4g c4p c4n2 c4n2g c4" || exit 1
cp gpurun_out/step_${TAG}_*.txt $OUT/ 2>/dev/null
