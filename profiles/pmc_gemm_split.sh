#!/bin/bash
# SQ counters of the split-fp16 matrix-core product (k_gemm_split_f16 / _w8 and its pre-pass): is the matrix pipe busy,
# and what do the waves wait for?  Separate rocprofv3 passes (counter slots), program directly after `--`,
# --kernel-trace only (the round-1 numbers existed only as scratch).
#   bash profiles/pmc_gemm_split.sh r04   ->  profiles/r04/pmc_gemm_split.txt
TAG=${1:-r04}
export TMPDIR=/tmp
OUT=gpurun_out/pmc_gemm
mkdir -p $OUT profiles/$TAG
CMD="python3 profiles/bench_gemm_split.py"
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/a -- $CMD > $OUT/a.log 2>&1 || { tail -5 $OUT/a.log; exit 1; }
rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES --kernel-trace --output-format csv -d $OUT/b -- $CMD > $OUT/b.log 2>&1 || { tail -5 $OUT/b.log; exit 1; }
rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/c -- $CMD > $OUT/c.log 2>&1 || { tail -5 $OUT/c.log; exit 1; }
python3 - > profiles/$TAG/pmc_gemm_split.txt <<'PY'
import csv, glob, collections
print("# rocprofv3 --pmc (three passes) of `python3 profiles/bench_gemm_split.py`: mean per launch, by kernel")
for d in ("a", "b", "c"):
    fs = glob.glob(f"gpurun_out/pmc_gemm/{d}/**/*_counter_collection.csv", recursive=True)
    if not fs:
        print(f"# pass {d}: no counter file"); continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        for tag in ("k_gemm_split_w8", "k_gemm_split_f16", "k_split_rows", "k_split_tile32", "k_gemm_f32_mfma"):
            if tag in k:
                agg[(tag, r["Counter_Name"])].append(float(r["Counter_Value"]))
                break
    for (k, c), v in sorted(agg.items()):
        print(f"{k:18s} {c:32s} n={len(v):3d} mean={sum(v)/len(v):.5g}")
PY
cat profiles/$TAG/pmc_gemm_split.txt
