"""Turn gpurun_out/prof_<tag>/ (tools/profile.sh) into committed evidence under profiles/:
   profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary (verbatim)
   profiles/<tag>_pmc.json           per-launch averages of every counter for the render kernel
   profiles/<tag>_summary.md         derived figures (VALU utilisation, occupancy, LDS, HBM bytes)
   profiles/hbm_traffic.json         HBM bytes per launch (FETCH_SIZE x2 on gfx950 + WRITE_SIZE), read by bench.py
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)

shutil.copy(os.path.join(src, "trace", "trace_kernel_stats.csv"), os.path.join(dst, f"{tag}_kernel_stats.csv"))
bench = json.loads(open(os.path.join(src, "trace_bench.json")).read().strip().splitlines()[-1])

pmc = {}
meta = {}
for f in sorted(glob.glob(os.path.join(src, "pmc*", "pmc*_counter_collection.csv"))):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "render_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta = {k: r[k] for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count")}
    for k, v in agg.items():
        pmc[k] = sum(v) / len(v)
json.dump({"kernel": "render_kernel<true>", "launches_averaged": 4, "dispatch": meta, "counters_per_launch": pmc}, open(os.path.join(dst, f"{tag}_pmc.json"), "w"), indent=1)

stats = {r["Name"]: r for r in csv.DictReader(open(os.path.join(src, "trace", "trace_kernel_stats.csv")))}
rk = next(v for k, v in stats.items() if "render_kernel" in k)
avg_ms = float(rk["AverageNs"]) / 1e6
n_simd = 256 * 4
rays = bench["roofline"]["rays_per_launch"]
wave_passes = rays / 64 / bench["roofline"]["lane_utilisation"]
# FETCH_SIZE / WRITE_SIZE are reported in KiB-like units of 1024 B by rocprofv3; gfx950: FETCH_SIZE counts 64 B per 128-B request -> x2
fetch_b = pmc.get("FETCH_SIZE", 0.0) * 1024 * 2
write_b = pmc.get("WRITE_SIZE", 0.0) * 1024
json.dump({"tag": tag, "kernel_tag": bench["roofline"].get("kernel_tag"), "bytes_per_launch": fetch_b + write_b, "fetch_bytes": fetch_b, "write_bytes": write_b,
           "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); WRITE_SIZE as reported; per render_kernel launch"},
          open(os.path.join(dst, "hbm_traffic.json"), "w"), indent=1)
cyc = pmc.get("GRBM_GUI_ACTIVE", 0.0) / 8          # summed over 8 XCDs
clock_ghz = cyc / (avg_ms * 1e-3) / 1e9 if cyc else float("nan")
valu = pmc.get("SQ_INSTS_VALU", 0.0)
lines = [
    f"# rocprofv3 summary `{tag}` — bench.py workload (1200x800, 500 spp, depth 50, 484 spheres), 1 x MI355X",
    "",
    f"Command: `rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline` (+ one `--pmc` run per counter group, tools/profile.sh).",
    f"bench line of the traced run: {bench['value']:.0f} Mrays/s, roofline.frac {bench['roofline']['frac']:.3f}, HIP-event kernel time {bench['roofline']['kernel_ms']:.2f} ms.",
    "",
    "| figure | value | how |",
    "|---|---|---|",
    f"| render_kernel average duration (kernel trace) | {avg_ms:.2f} ms over {rk['Calls']} calls ({rk['Percentage']} % of GPU time) | `{tag}_kernel_stats.csv` |",
    f"| dispatch | grid {meta.get('Grid_Size')} threads, workgroup {meta.get('Workgroup_Size')}, LDS {meta.get('LDS_Block_Size')} B, VGPR {meta.get('VGPR_Count')}, SGPR {meta.get('SGPR_Count')}, scratch {meta.get('Scratch_Size')} | counter CSV |",
    f"| waves per launch / per SIMD (occupancy) | {pmc.get('SQ_WAVES', 0):.0f} / {pmc.get('SQ_WAVES', 0) / n_simd:.1f} of 8 | SQ_WAVES (persistent grid: all resident for the whole kernel) |",
    f"| effective clock | {clock_ghz:.2f} GHz | GRBM_GUI_ACTIVE / 8 / duration |",
    f"| VALU wave-instructions per launch | {valu:.3e} = {valu / wave_passes:.0f} per wavefront pass ({484 * 11} of them are the 11-instruction filter over 484 spheres) | SQ_INSTS_VALU |",
    f"| VALU issue utilisation | {valu / n_simd * 2 / cyc:.3f} (2 cycles per wave64 VALU instruction per SIMD) | SQ_INSTS_VALU x 2 / (SIMDs x cycles) |",
    f"| FP32 FMA / ADD / MUL / transcendental instructions | {pmc.get('SQ_INSTS_VALU_FMA_F32', 0):.3e} / {pmc.get('SQ_INSTS_VALU_ADD_F32', 0):.3e} / {pmc.get('SQ_INSTS_VALU_MUL_F32', 0):.3e} / {pmc.get('SQ_INSTS_VALU_TRANS_F32', 0):.3e} | SQ_INSTS_VALU_* |",
    f"| executed fp32 flops (2 per FMA, 1 per ADD/MUL) as a fraction of the 157.3 TF peak | {(2 * pmc.get('SQ_INSTS_VALU_FMA_F32', 0) + pmc.get('SQ_INSTS_VALU_ADD_F32', 0) + pmc.get('SQ_INSTS_VALU_MUL_F32', 0)) * 64 / (avg_ms * 1e-3) / 157.3e12:.3f} | counters x 64 lanes / duration |",
    f"| SALU / LDS / SMEM / VMEM wave-instructions | {pmc.get('SQ_INSTS_SALU', 0):.3e} / {pmc.get('SQ_INSTS_LDS', 0):.3e} / {pmc.get('SQ_INSTS_SMEM', 0):.3e} / {pmc.get('SQ_INSTS_VMEM', 0):.3e} | SQ_INSTS_* |",
    f"| LDS busy | {pmc.get('SQ_LDS_IDX_ACTIVE', 0) / 256 / cyc:.3f} of kernel cycles per CU; bank-conflict cycles {pmc.get('SQ_LDS_BANK_CONFLICT', 0) / max(1.0, pmc.get('SQ_LDS_IDX_ACTIVE', 1)):.4f} of LDS cycles | SQ_LDS_IDX_ACTIVE, SQ_LDS_BANK_CONFLICT |",
    f"| wave time split | active {pmc.get('SQ_ACTIVE_INST_ANY', 0) / max(1.0, pmc.get('SQ_WAVE_CYCLES', 1)):.3f}, waiting on issue {pmc.get('SQ_WAIT_INST_ANY', 0) / max(1.0, pmc.get('SQ_WAVE_CYCLES', 1)):.3f}, parked (waitcnt) {pmc.get('SQ_WAIT_ANY', 0) / max(1.0, pmc.get('SQ_WAVE_CYCLES', 1)):.3f} of SQ_WAVE_CYCLES | SQ_ACTIVE_INST_ANY, SQ_WAIT_INST_ANY, SQ_WAIT_ANY |",
    f"| HBM bytes per launch | fetch {fetch_b / 1e6:.2f} MB + write {write_b / 1e6:.1f} MB = {(fetch_b + write_b) / (avg_ms * 1e-3) / 1e9:.1f} GB/s ({(fetch_b + write_b) / (avg_ms * 1e-3) / 8e12 * 100:.2f} % of 8 TB/s) | FETCH_SIZE x 2 (gfx950 correction), WRITE_SIZE |",
    f"| L2 | {pmc.get('TCC_REQ_sum', 0):.3e} requests, {pmc.get('TCC_ATOMIC_sum', 0):.3e} atomics (the fixed-point pixel accumulators) | TCC_* |",
    "",
    "Raw per-launch counter averages: `%s_pmc.json`." % tag,
]
open(os.path.join(dst, f"{tag}_summary.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
