#!/bin/bash
# SQ counters of the shared-negative L1 kernels (C4 shapes): where do the wave cycles go?
export TMPDIR=/tmp
OUT=gpurun_out/pmc_l1
mkdir -p $OUT
CMD="python3 profiles/microbench.py 4096x4352"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/a -- $CMD > $OUT/a.log 2>&1 || { tail -5 $OUT/a.log; exit 1; }
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT/b -- $CMD > $OUT/b.log 2>&1 || { tail -5 $OUT/b.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections
for d in ("a", "b"):
    f = glob.glob(f"gpurun_out/pmc_l1/{d}/**/*_counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_neg_shared_bwd" in k or "k_l1_fwd_pk" in k or "k_l1_bwd_both" in k:
            short = "fwd_pk" if "k_l1_fwd" in k else ("bwd_both" if "k_l1_bwd_both" in k else "bwd")  # one launch for both products
            agg[(short, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(agg.items()):
        print(f"{k:8s} {c:24s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
PY
