#!/usr/bin/env python3
"""Turn gpurun_out/prof_<tag>/ (written by profiles/run_profiles.sh on the GPU box)
into the tracked summary profiles/<tag>/SUMMARY.md + profiles/pmc_traffic.json.

    python profiles/summarize.py r01
"""
import collections
import csv
import glob
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = f"gpurun_out/prof_{tag}"
out = f"profiles/{tag}"
os.makedirs(out, exist_ok=True)
lines = []
for d, title in (("score", "bench.py --steps 20 --warmup 5 --no-cpu-baseline   (default: score mode)"),
                 ("train", "bench.py --steps 20 --warmup 5 --no-cpu-baseline --mode train")):
    f = max(glob.glob(f"{src}/{d}/runc/*_kernel_stats.csv"), key=os.path.getmtime)
    lines.append(f"## rocprofv3 --kernel-trace --stats -- python3 {title}\n\n")
    lines.append("| kernel | calls | avg (us) | min (us) | max (us) | % of GPU time |\n|---|---|---|---|---|---|\n")
    for r in list(csv.DictReader(open(f)))[:18]:
        lines.append(f"| `{r['Name'][:120]}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | {float(r['MinNs'])/1e3:.1f} |"
                     f" {float(r['MaxNs'])/1e3:.1f} | {float(r['Percentage']):.2f} |\n")
    lines.append("\n")
    os.system(f"cp {f} {out}/{d}_kernel_stats.csv")
pm = {}
for d in ("pmc_fetch", "pmc_write", "pmc_fetch_hbm"):
    f = max(glob.glob(f"{src}/{d}/runc/*_counter_collection.csv"), key=os.path.getmtime)
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if "k_neg_pertriple_fwd" in k[0] and "false>" in k[0]:  # the plain forward, not the fused training variant
            pm[d] = (k[1], len(v), sum(v) / len(v))
fetch, write, fetch_hbm = pm["pmc_fetch"][2], pm["pmc_write"][2], pm["pmc_fetch_hbm"][2]
lines.append("## rocprofv3 --pmc (separate passes), kernel k_neg_pertriple_fwd<float, 4, 8, DOT, 2>\n\n")
lines.append("| pass | counter | dispatches | mean value (KiB) | bytes per launch after the gfx950 correction |\n|---|---|---|---|---|\n")
lines.append(f"| 93,773-row shard (192 MB, Infinity-Cache resident) | FETCH_SIZE | {pm['pmc_fetch'][1]} | {fetch:.1f} | 2 x {fetch:.0f} x 1024 = {2*fetch*1024:.4g} |\n")
lines.append(f"| same | WRITE_SIZE | {pm['pmc_write'][1]} | {write:.1f} | {write*1024:.4g} |\n")
lines.append(f"| 4,000,000-row shard (8.2 GB, HBM resident) | FETCH_SIZE | {pm['pmc_fetch_hbm'][1]} | {fetch_hbm:.1f} | 2 x {fetch_hbm:.0f} x 1024 = {2*fetch_hbm*1024:.4g} |\n")
lines.append("\nFETCH_SIZE is doubled as MI355X_MICROARCH.md (HBM section) prescribes for 16 B/lane coalesced reads on gfx950;"
             " WRITE_SIZE is used as read (4-byte scattered score stores: uncalibrated width, small).  Algorithmic bytes"
             " of the launch: 2,164,260,864.\n")
pmc = {"neg_score_pertriple_fwd_bytes_per_launch": 2 * fetch * 1024 + write * 1024, "fetch_size_kib": fetch,
       "write_size_kib": write, "fetch_size_kib_hbm_resident_8GB_shard": fetch_hbm,
       "hbm_resident_bytes_per_launch": 2 * fetch_hbm * 1024 + write * 1024,
       "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of `python3 bench.py --steps 20 --warmup 5"
               " --no-cpu-baseline --no-extra-legs`; traffic = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (profiles/summarize.py)"}
json.dump(pmc, open(f"{out}/pmc_traffic.json", "w"), indent=1)
json.dump(pmc, open("profiles/pmc_traffic.json", "w"), indent=1)
for name in ("score.json", "train.json", "score_hbm.json"):
    os.system(f"cp {src}/{name} {out}/bench_{name}")
os.system(f"cp {src}/step_{tag}_*.txt {out}/ 2>/dev/null")
steps = sorted(glob.glob(f"{out}/step_{tag}_*.txt"))
if steps:
    lines.append("## One training step, kernel by kernel (`profiles/step_trace.py`, rocprofv3 --kernel-trace)\n\n"
                 "| file | workload | dispatches | span (us) | sum of kernel durations (us) |\n|---|---|---|---|---|\n")
    what = {"c2": "C2 ComplEx d=256 fp32, S=4096 x 256 per-triple, SGD", "c2adam": "same, AdamW",
            "c2score": "C2, the bench headline step (gather + score + loss)",
            "c2sm": "C2 training step in its multi-GPU form (ScoreMovingBessKGE, fused forward with partials), one shard",
            "c4s": "C4 TransE d=256 fp16, S=512, K=32, eager",
            "c4n2": "C4, two shards of S=512 stepped in lock-step on one GPU (the n > 1 code path: pack, exchange, C8), eager", "c4g": "C4 TransE d=256 fp16, S=512, K=32, hipGraph replay",
            "c4p": "the same replayed from a C-side step plan (`Options.use_plans`, `bess_plan_run`)",
            "c4n2g": "C4, two shards of S=512 in lock-step on one GPU, replayed from a hipGraph",
            "c4": "C4 TransE d=256 fp16, S=4096, K=256, eager",
            "c2em2": "C2's scorer and per-triple negatives in the EmbeddingMoving form on TWO shards stepped in lock-step on one GPU (S = 2048, 2 x 128 negatives per triple and shard; the n > 1 code path: pack, exchange - simulated by copies here -, fused forward over the received rows, in-place negative gradients, C8, coalesced update), SGD"}
    for f in steps:
        txt = open(f).read().splitlines()
        key = os.path.basename(f)[len(f"step_{tag}_"):-4]
        n = txt[0].split()[1]
        tail = txt[-1].replace("# span ", "").replace(" us, sum of kernel durations ", "|").replace(" us", "")
        span, busy = tail.split("|")
        lines.append(f"| `{os.path.basename(f)}` | {what.get(key, key)} | {n} | {span} | {busy} |\n")
    lines.append("\n")
# index of the other logs kept for this round (each is the stdout of the script named in its first column)
INDEX = [
    ("bench_default.json", "python bench.py", "the driver's line: value, roofline, train_step (SGD and AdamW), cpu_baseline"),
    ("bench_c4_4096x256.json", "python bench.py --workload c4", "north_star's scaling workload as the line itself (S = 4096, K = 256 per shard pair)"),
    ("bench_c4_512x32.json", "python bench.py --workload c4 --c4-point 512,32 --steps 256 --warmup 32", "the same at the notebook's micro-batch"),
    ("pmc_k9_r03.txt", "profiles/pmc_k9.sh", "counters of the K9 segmented reduction inside the C2 training step (L2 hit rate, fetch bytes, VALU / wait cycles)"),
    ("microbench_final.log", "profiles/microbench.py", "every hot entry point on the BASELINE shapes (GB/s, TFLOP/s, T lane-ops/s)"),
    ("bench_gemm_split.log", "profiles/bench_gemm_split.py", "split-fp16 matrix-core products: accuracy vs float64 and rate, forward + backward"),
    ("bench_topk.log", "profiles/bench_topk.py", "top-k over all entities (YAGO3-10, wikikg2, biokg shapes) and full ranks of the same queries: counted in the scoring epilogue vs score matrix + `bess_ranks_from_scores`"),
    ("sweep_l1_bwd.log", "profiles/sweep_l1_bwd.py", "`k_l1_bwd_both` (atomic sums) and `k_l1_bwd_parts` (partial sums) alone (hipGraph of back-to-back launches) from 256 x 288 to 8192 x 8448"),
    ("pmc_l1_kernels.txt", "profiles/pmc_l1_bwd.sh / pmc_l1_r04.sh", "SQ counters of the shared-negative L1 kernels (r04: 512 x 544, 2048 x 2176, 8192 x 8448)"),
    ("pmc_gemm_split.txt", "profiles/pmc_gemm_split.sh", "SQ counters of the split-fp16 GEMM and its pre-pass; reading: `pmc_gemm_split.md`"),
    ("gemm_kernel_trace.txt", "rocprofv3 --kernel-trace -- python3 profiles/bench_gemm_split.py", "un-profiled durations of `k_gemm_split_w8` and its pre-pass per shape"),
    ("bench_tail.log", "profiles/bench_tail.py", "`bess_pertriple_tail` against the four launches it replaces, C2 and notebook shapes"),
    ("timeline_r04_c2.txt", "profiles/step_timeline.py c2", "HIP-event timeline of a C2 training step, streams not serialised: the index build under the forward"),
    ("timeline_r04_c2adam.txt", "profiles/step_timeline.py c2adam", "the same with AdamW"),
    ("bench_index.log", "profiles/bench_index.py", "segment-index build: the library's kernel-only radix sort against rocPRIM's"),
    ("graph_memset_probe.md", "profiles/graph_memset_probe.py", "what memset nodes of recorded steps do on replay (probe, DOT dump, fault tail)"),
    ("bench_c4.log", "profiles/bench_c4.py", "BASELINE configs[3] regime through runtime.Runner, hipGraph replay, SGD and AdamW"),
    ("bench_graphs.log", "profiles/bench_graphs.py", "hipGraph replay in the launch-bound regime (configs[0] shape)"),
    ("bench_optim.log", "profiles/bench_optim.py", "training step of the bench workload per optimiser"),
    ("bench_skew.log", "profiles/bench_skew.py", "segmented K9 when a share of all references points at one row"),
    ("bench_sampler.log", "profiles/bench_sampler.py", "device-side samplers vs numpy"),
    ("stress_long_rows.log", "profiles/stress_long_rows.py", "400 random skewed problems through the long-row tier of K9"),
    ("stress_gemm_split.log", "profiles/stress_gemm_split.py", "80 random shapes through the split-fp16 products vs float64"),
    ("stress_pertriple.log", "profiles/stress_pertriple.py", "150 random problems: dominant kernel and its backward vs torch float64"),
    ("stress_fused_forward.log", "profiles/stress_fused_forward.py", "160 random problems: fused training forward vs the two-pass path"),
    ("stress_shared_distance.log", "profiles/stress_shared_distance.py", "120 random shapes through the shared-negative L1 / L2 kernels (all tile variants, reduction splits, one-launch backward, packed fp16 forward) vs float64"),
    ("stress_topk.log", "profiles/stress_topk.py", "150 random top-k problems (ties, masks, padded rows) vs a stable sort"),
    ("ubench_hbm_bw.log", "profiles/ubench/hbm_bw.hip", "what the memory system delivers: streaming and random-row reads"),
    ("ubench_valu_rate.log", "profiles/ubench/valu_rate.hip", "VALU ceiling of the p-norm tile kernels (scalar and packed)"),
    ("ubench_valu_pk.log", "profiles/ubench/valu_pk.hip", "issue rate of every packed 16-bit / dot2 instruction against v_add_f32 (0.6x), and of the packed L1 forms"),
    ("bench_c5.log", "profiles/bench_c5.py", "BASELINE configs[4]: one shard at its real size (62.5 M x 512 fp32 = 128 GB), dominant kernel TB/s, train step"),
    ("microbench_l1.log", "profiles/microbench.py \"L1 shared\"", "packed-fp16 L1 forward vs the fp32 kernels; backward"),
    ("ubench_mfma_f32.log", "profiles/ubench/mfma_f32.hip", "fp32 MFMA inner loop ceiling"),
    ("ubench_mfma_f16.log", "profiles/ubench/mfma_f16.hip", "fp16 MFMA consumer loop of the split GEMM: rate and clocks"),
    ("ubench_l1_tile.log", "profiles/ubench/l1_tile.hip", "the steps that took the L1 tile kernel from 35 to 48 T lane-ops/s"),
]
lines.append("\n## Other logs of this round\n\n| file | produced by | what |\n|---|---|---|\n")
for name, prod, what in INDEX:
    if os.path.exists(f"{out}/{name}"):
        lines.append(f"| `{name}` | `{prod}` | {what} |\n")
open(f"{out}/SUMMARY.md", "w").write(
    f"# Round {tag} profiles (MI355X, ROCm 7.2, rocprofv3)\n\nRecipe: `profiles/run_profiles.sh {tag}` on the GPU box (gpurun),"
    f" then `python profiles/summarize.py {tag}`; the raw stats CSVs and the bench lines of the profiled runs are next to"
    " this file.\n\n" + "".join(lines))
print(open(f"{out}/SUMMARY.md").read())
