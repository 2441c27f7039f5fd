#!/bin/bash
# SQ counters of the shared-negative L1 kernels at the C4 shape (k_l1_fwd_pk, k_l1_bwd_both, k_l1_bwd_parts at the
# notebook shape): bank conflicts of the packed forward, what the backward's 3.2 cycles per instruction wait on.
#   bash profiles/pmc_l1_r04.sh r04  ->  profiles/r04/pmc_l1_kernels.txt
TAG=${1:-r04}
export TMPDIR=/tmp
OUT=gpurun_out/pmc_l1
mkdir -p $OUT profiles/$TAG
CMD="python3 profiles/pmc_l1_driver.py"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/a -- $CMD > $OUT/a.log 2>&1 || { tail -5 $OUT/a.log; exit 1; }
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_WAVES SQ_INSTS_SMEM SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT/b -- $CMD > $OUT/b.log 2>&1 || { tail -5 $OUT/b.log; exit 1; }
rocprofv3 --pmc SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL --kernel-trace --output-format csv -d $OUT/c -- $CMD > $OUT/c.log 2>&1 || { tail -5 $OUT/c.log; }
python3 - > profiles/$TAG/pmc_l1_kernels.txt <<'PY'
import csv, glob, collections
print("# rocprofv3 --pmc of `python3 profiles/pmc_l1_driver.py` (S x N = 512 x 544, 2048 x 2176, 8192 x 8448; W = 256 fp16): mean per launch, by kernel and grid")
for d in ("a", "b", "c"):
    fs = glob.glob(f"gpurun_out/pmc_l1/{d}/**/*_counter_collection.csv", recursive=True)
    if not fs:
        print(f"# pass {d}: no counter file"); continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        for tag in ("k_l1_fwd_pk", "k_l1_bwd_both", "k_l1_bwd_parts", "k_neg_shared_bwd"):
            if tag in k:
                grid = r.get("Grid_Size", "")
                agg[(tag, grid, r["Counter_Name"])].append(float(r["Counter_Value"]))
                break
    for (k, g, c), v in sorted(agg.items()):
        print(f"{k:16s} grid={g:>9s} {c:26s} n={len(v):3d} mean={sum(v)/len(v):.5g}")
PY
head -60 profiles/$TAG/pmc_l1_kernels.txt
