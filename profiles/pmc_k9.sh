#!/bin/bash
# Counters of the K9 segmented reduction (k_pertriple_grad_segments) inside the C2 training step:
#   bash profiles/pmc_k9.sh r03      -> gpurun_out/pmc_k9_<tag>.txt
# Separate rocprofv3 passes (counter slots), program after `--`, --kernel-trace only.
set -o pipefail
TAG=${1:-r03}
export TMPDIR=/tmp
OUT=gpurun_out/pmc_k9_$TAG
mkdir -p $OUT
CMD="python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs --mode train"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU" \
           "SQ_INST_CYCLES_VMEM SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_SMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -- $CMD > $OUT/p$i.log 2>&1 || { echo "pass $i ($set) failed"; tail -3 $OUT/p$i.log; }
done
python3 - "$OUT" <<'PY' | tee gpurun_out/pmc_k9_$TAG.txt
import csv, glob, collections, sys
out = sys.argv[1]
agg = collections.defaultdict(list)
dur = collections.defaultdict(list)
for f in glob.glob(f"{out}/p*/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        for short in ("k_pertriple_grad_segments", "k_neg_pertriple_fwd", "k_long_segments"):
            if short in k:
                agg[(short, r["Counter_Name"])].append(float(r["Counter_Value"]))
for f in glob.glob(f"{out}/p1/**/*_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        for short in ("k_pertriple_grad_segments", "k_neg_pertriple_fwd"):
            if short in r["Kernel_Name"]:
                dur[short].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("# rocprofv3 --pmc (separate passes) of `python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs --mode train`")
print("# per dispatch means; k_pertriple_grad_segments runs once per column window (2 per step)")
for k, v in sorted(dur.items()):
    print(f"{k:28s} duration_us                n={len(v):3d} mean={sum(v)/len(v):.1f}")
for (k, c), v in sorted(agg.items()):
    print(f"{k:28s} {c:24s} n={len(v):3d} mean={sum(v)/len(v):.5g}")
PY
