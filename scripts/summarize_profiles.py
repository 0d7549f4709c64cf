#!/usr/bin/env python3
"""Copies one collection (gpurun_out/<tag>_* written by scripts/collect_profiles.sh) into profiles/ and derives the two summaries
that need arithmetic: the frame-shaped launches of the dominant kernel in the profiled bench run, and the PMC ratios.
    python scripts/summarize_profiles.py r02"""
import collections
import csv
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
src, dst = "gpurun_out/", "profiles/"
KERNEL = "k_encode_mlp_f16x3_t128<false, false, false>"          # the headline kernel since round 6 (rounds 2-5: k_encode_mlp_f16x3_dual<false, false, false>)

rows = [r for r in csv.DictReader(open(f"{src}prof/{tag}_bench/bench_kernel_trace.csv")) if KERNEL in r["Kernel_Name"]]
durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
coarse = [d for d in durs if 100 < d < 200]
fine = [d for d in durs if d >= 200]
prof = json.load(open(f"{src}{tag}_bench_under_rocprof.json"))
plain = json.load(open(f"{src}{tag}_bench.json"))
mean = lambda v: sum(v) / len(v)
with open(f"{dst}{tag}_bench_dual_launches.txt", "w") as f:
    f.write(f"# {KERNEL} launches in the profiled bench.py run (rocprofv3 --kernel-trace; rows in {tag}_bench_kernel_trace_dual.csv)\n")
    f.write(f"# {len(durs)} launches in total; the run also launches the kernel at other shapes (80 000-ray band, coarse-only leg, warm-ups)\n")
    f.write(f"frame-shaped launches (640 000 rays): coarse x64 samples: n={len(coarse)} mean {mean(coarse):.2f} ms; fine x192 samples: n={len(fine)} mean {mean(fine):.2f} ms\n")
    f.write(f"mean of the coarse and fine launch = {(mean(coarse) + mean(fine)) / 2:.2f} ms   (bench.py's HIP-event figure in the same run: "
            f"{prof['roofline']['avg_launch_ms']:.2f} ms; un-profiled run: {plain['roofline']['avg_launch_ms']:.2f} ms)\n")
with open(f"{dst}{tag}_bench_kernel_trace_dual.csv", "w") as f:
    w = csv.writer(f)
    w.writerow(["Kernel_Name", "Grid_Size_X", "Start_Timestamp", "End_Timestamp", "Duration_ms"])
    for r, d in zip(rows, durs):
        w.writerow([r["Kernel_Name"], r.get("Grid_Size_X", ""), r["Start_Timestamp"], r["End_Timestamp"], f"{d:.3f}"])
for a, b in (("bench.json", "bench.json"), ("bench_under_rocprof.json", "bench_under_rocprof.json"), ("bench_kernel_stats.csv", "bench_kernel_stats.csv"),
             ("mlp_pmc_dual_A.csv", "mlp_pmc_dual_passA.csv"), ("mlp_pmc_dual_fetch.csv", "mlp_pmc_dual_fetch.csv"),
             ("mlp_pmc_dual_write.csv", "mlp_pmc_dual_write.csv"), ("mlp_pmc_dual_cache.csv", "mlp_pmc_dual_cache.csv"),
             ("mlp_pmc_summary.txt", "mlp_pmc_summary.txt"), ("kernel_forms.txt", "kernel_forms_same_box.txt"), ("ssr_frame.txt", "ssr_frame.txt"),
             ("train_step.txt", "train_step.txt"), ("train_step_kernel_stats.csv", "train_step_kernel_stats.csv"),
             ("train_pmc_summary.txt", "train_pmc_summary.txt"), ("train_pmc_w.csv", "train_pmc_w.csv"), ("train_pmc_f.csv", "train_pmc_f.csv"),
             ("train_pmc_m.csv", "train_pmc_m.csv"), ("train_kernels.txt", "train_kernels.txt"), ("bench_n2_shared.json", "bench_n2_shared.json"),
             ("trained_network.txt", "trained_network.txt")):
    if os.path.exists(f"{src}{tag}_{a}"):
        shutil.copy(f"{src}{tag}_{a}", f"{dst}{tag}_{b}")


def pmc(path):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        agg[(r["Kernel_Name"].split("(")[0][-64:], r["Counter_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


print(open(f"{dst}{tag}_bench_dual_launches.txt").read())
ours = lambda d: {k: v for k, v in d.items() if "inerf::" in k[0]}
lines = []
a = ours(pmc(f"{dst}{tag}_mlp_pmc_dual_passA.csv"))
pts = 640000 * 192
for (k, c), v in sorted(a.items()):
    if c == "SQ_VALU_MFMA_BUSY_CYCLES":
        cyc = a[(k, "GRBM_GUI_ACTIVE")] / 8
        lines.append(f"{k.strip()}: {cyc:.4g} cycles per launch, MFMA busy {v / 1024 / cyc * 100:.1f} %, "
                     f"waves resident per SIMD {a[(k, 'SQ_WAVE_CYCLES')] * 4 / 1024 / cyc:.2f}")
t = {}
for f in ("fetch", "write", "cache"):
    for (k, c), v in ours(pmc(f"{dst}{tag}_mlp_pmc_dual_{f}.csv")).items():
        t[c] = v
rd, wr = t["FETCH_SIZE"] * 1024 / 1e6, t["WRITE_SIZE"] * 1024 / 1e6
alg = pts * 48.0 + 640000 * 44 + 2.7e6
lines.append(f"bench frame's fine launch ({pts} points): FETCH_SIZE {rd:.1f} MB + WRITE_SIZE {wr:.1f} MB = {rd + wr:.1f} MB = {(rd + wr) * 1e6 / pts:.1f} B per point "
             f"= {(rd + wr) * 1e6 / alg:.2f} x algorithmic ({alg / 1e6:.1f} MB); L2 hit rate {t['TCC_HIT_sum'] / (t['TCC_HIT_sum'] + t['TCC_MISS_sum']) * 100:.2f} %")
m = ours(pmc(f"{dst}{tag}_train_pmc_m.csv"))
w = ours(pmc(f"{dst}{tag}_train_pmc_w.csv"))
fz = ours(pmc(f"{dst}{tag}_train_pmc_f.csv"))
lines.append("training kernels on the fine-pass batch (2048 rays x 192 = 393 216 points), per launch:")
for (k, c), v in sorted(m.items()):
    if c == "SQ_VALU_MFMA_BUSY_CYCLES":
        lines.append(f"  {k.strip()}: MFMA busy {v / 1024 / (m[(k, 'GRBM_GUI_ACTIVE')] / 8) * 100:.1f} %, WRITE_SIZE {w[(k, 'WRITE_SIZE')] * 1024 / 1e9:.3f} GB, "
                     f"FETCH_SIZE {fz[(k, 'FETCH_SIZE')] * 1024 / 1e9:.3f} GB (uncorrected: gfx950 reports half of 16-byte-per-lane streams)")
open(f"{dst}{tag}_pmc_digest.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
