"""Turn gpurun_out/prof_<tag>/ (tools/profile.sh) into committed evidence under profiles/:
   profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary (verbatim)
   profiles/<tag>_pmc.json           per-launch averages of every counter for the render kernel
   profiles/<tag>_summary.md         derived figures (VALU utilisation, occupancy, LDS, HBM bytes)
   profiles/hbm_traffic.json         HBM bytes per launch (FETCH_SIZE x2 on gfx950 + WRITE_SIZE), read by bench.py

    python tools/summarize_profile.py <tag> [path/to/librtk.so] [LDS bytes per workgroup, if the bench line lacks it]

Everything kernel-specific comes from the bench line of the traced run (kernel name, kernel tag, filter instruction
count, sphere count, rays per launch) and from the code object of the library that was profiled (registers, LDS,
scratch: tools/code_object_info.py) — rocprofv3's per-dispatch VGPR/LDS columns are not used (they were wrong in round 1).
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from code_object_info import kernel_info  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
lib_path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "raytracinginoneweekend_amd", "librtk.so")
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
PEAK_TF = 157.3
N_SIMD = 256 * 4

shutil.copy(os.path.join(src, "trace", "trace_kernel_stats.csv"), os.path.join(dst, f"{tag}_kernel_stats.csv"))
bench = json.loads(open(os.path.join(src, "trace_bench.json")).read().strip().splitlines()[-1])
roof = bench["roofline"]
kernel = roof["kernel"]                                   # e.g. "render_kernel<true>"
kernel_key = kernel.split("<")[0]
n_spheres = int(round(roof["flop_per_ray"] / 17))
filt_instr = roof.get("filter_valu_per_test")

# ---- counters: per-launch averages over every launch of the render kernel in each pass
pmc, launches, grid = {}, {}, {}
for f in sorted(glob.glob(os.path.join(src, "pmc*", "pmc*_counter_collection.csv"))):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if kernel_key in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            grid = {"Grid_Size": r["Grid_Size"], "Workgroup_Size": r["Workgroup_Size"]}
    for k, v in agg.items():
        pmc[k] = sum(v) / len(v)
        launches[k] = len(v)

# ---- the code object of the profiled library
co = kernel_info(lib_path)
# the 4-wave instantiation (the 8-wave one serves large general-word scenes only)
want = "render_kernel<true, false, 4>" if kernel == "render_kernel<true>" else "render_kernel<false, false, 4>" if kernel == "render_kernel<false>" else kernel
ck = next((v for k, v in co.items() if want in k), None)
if ck is None:
    raise SystemExit(f"{want} not found in {lib_path}: {list(co)}")
lds_dynamic = roof.get("lds_bytes_per_workgroup") or (int(sys.argv[3]) if len(sys.argv) > 3 else None)   # bench line (or 3rd argument): static + dynamic bytes the launch asks for
lds_wg = lds_dynamic if lds_dynamic else ck["group_segment_fixed_size"]
waves_by_lds = (160 * 1024 // lds_wg) if lds_wg else 8    # one workgroup = 4 waves = one wave per SIMD
waves_by_vgpr = ck["waves_per_simd_by_vgpr"]

json.dump({"kernel": kernel, "kernel_tag": roof.get("kernel_tag"), "launches_averaged": launches, "dispatch": grid, "code_object": ck,
           "counters_per_launch": pmc}, open(os.path.join(dst, f"{tag}_pmc.json"), "w"), indent=1)

stats = {r["Name"]: r for r in csv.DictReader(open(os.path.join(src, "trace", "trace_kernel_stats.csv")))}
rk = next(v for k, v in stats.items() if kernel_key in k)
avg_ms = float(rk["AverageNs"]) / 1e6
rays = roof["rays_per_launch"]
wave_passes = rays / 64 / roof["lane_utilisation"]
alg_flop = rays * roof["flop_per_ray"]
ROOF_PEAK = roof.get('peak', PEAK_TF)
frac_trace = alg_flop / (avg_ms * 1e-3) / 1e12 / ROOF_PEAK
frac_trace_vec = alg_flop / (avg_ms * 1e-3) / 1e12 / PEAK_TF
# FETCH_SIZE / WRITE_SIZE are in units of 1024 B; gfx950: FETCH_SIZE tallies 128-B requests at 64 B -> x2 (MI355X_MICROARCH.md, HBM)
fetch_b = pmc.get("FETCH_SIZE", 0.0) * 1024 * 2
write_b = pmc.get("WRITE_SIZE", 0.0) * 1024
have_traffic = "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc
cyc = pmc.get("GRBM_GUI_ACTIVE", 0.0) / 8          # summed over 8 XCDs
clock_ghz = cyc / (avg_ms * 1e-3) / 1e9 if cyc else float("nan")
valu = pmc.get("SQ_INSTS_VALU", 0.0)
mfma_busy_c = pmc.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
coexec_c = pmc.get("SQ_VALU_MFMA_COEXEC_CYCLES")
valu_util = valu / N_SIMD * 2 / cyc if cyc else None
mfma_busy = mfma_busy_c / N_SIMD / cyc if cyc else None
coexec = coexec_c / N_SIMD / cyc if (cyc and coexec_c is not None) else None
if have_traffic:
    json.dump({"tag": tag, "kernel_tag": roof.get("kernel_tag"), "bytes_per_launch": fetch_b + write_b, "fetch_bytes": fetch_b, "write_bytes": write_b,
               "algorithmic_bytes": roof.get("algorithmic_bytes_per_launch"),
               "valu_issue_util": valu_util, "mfma_busy": mfma_busy, "mfma_valu_coexec": coexec,
               "source": f"profiles/{tag}_summary.md (rocprofv3 --pmc of bench.py's headline workload, one GPU; fractions of SIMD cycles)",
               "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); WRITE_SIZE as reported "
                       "(calibrated there for dword float atomics; this kernel's atomics are 8-byte integer adds); per render_kernel launch"},
              open(os.path.join(dst, "hbm_traffic.json"), "w"), indent=1)
fma, add, mul, trans = (pmc.get("SQ_INSTS_VALU_" + k, 0.0) for k in ("FMA_F32", "ADD_F32", "MUL_F32", "TRANS_F32"))
exec_frac = (2 * fma + add + mul) * 64 / (avg_ms * 1e-3) / (PEAK_TF * 1e12)
wc = max(1.0, pmc.get("SQ_WAVE_CYCLES", 1))
g = lambda k: pmc.get(k, 0.0)   # noqa: E731
filt_txt = (f"{n_spheres * filt_instr:.0f} of them are the {filt_instr}-instruction filter over {n_spheres} spheres = "
            f"{n_spheres * filt_instr / (valu / wave_passes) * 100:.0f} %") if filt_instr and valu else "filter share: n/a"
lines = [
    f"# rocprofv3 summary `{tag}` — {bench['config']['workload']}, 1 x MI355X",
    "",
    f"Kernel `{kernel}`, tag `{roof.get('kernel_tag')}`.",
    "Command: `rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline` (+ one `--pmc` run per counter group, tools/profile.sh).",
    f"bench line of the traced run: {bench['value']:.0f} Mrays/s, roofline.frac {roof['frac']:.3f}, HIP-event kernel time {roof['kernel_ms']:.2f} ms.",
    "",
    "| figure | value | how |",
    "|---|---|---|",
    f"| render_kernel average duration (kernel trace) | {avg_ms:.2f} ms over {rk['Calls']} calls ({rk['Percentage']} % of GPU time) | `{tag}_kernel_stats.csv` |",
    f"| roofline fraction from the trace alone | {alg_flop:.4e} algorithmic flop per launch ({rays:.4e} rays x {roof['flop_per_ray']}) / {avg_ms:.2f} ms / {ROOF_PEAK:.1f} TF ({roof.get('bound')}) = **{frac_trace:.3f}** (bench line, HIP events: {roof['frac']:.3f}); against the {PEAK_TF} TF fp32 vector peak, the measure of rounds 1-2: {frac_trace_vec:.3f} | kernel trace + ray count of the traced run |",
    f"| launch | grid {grid.get('Grid_Size')} threads, workgroup {grid.get('Workgroup_Size')} | counter CSV |",
    f"| registers / LDS / scratch (code object) | {ck['vgpr_count']} VGPR + {ck['agpr_count']} AGPR, {ck['sgpr_count']} SGPR, spills {ck['vgpr_spill_count']} VGPR / {ck['sgpr_spill_count']} SGPR, static LDS {ck['group_segment_fixed_size']} B, "
    f"LDS per workgroup at launch {lds_wg} B, scratch {ck['private_segment_fixed_size']} B | `llvm-readelf --notes` of the profiled library (tools/code_object_info.py) |",
    f"| occupancy limits (waves per SIMD) | by VGPRs {waves_by_vgpr}, by LDS {min(8, waves_by_lds)} -> {min(waves_by_vgpr, waves_by_lds, 8)}; measured {g('SQ_WAVES') / N_SIMD:.1f} ({g('SQ_WAVES'):.0f} waves per launch, persistent: all resident for the whole kernel) | code object, SQ_WAVES |",
    f"| effective clock | {clock_ghz:.2f} GHz | GRBM_GUI_ACTIVE / 8 / duration |",
    f"| VALU wave-instructions per launch | {valu:.3e} = {valu / wave_passes:.0f} per wavefront pass ({filt_txt}) | SQ_INSTS_VALU, passes = rays / 64 / lane utilisation ({roof['lane_utilisation']:.4f}) |",
    f"| VALU issue utilisation | {valu / N_SIMD * 2 / cyc if cyc else float('nan'):.3f} if every vector instruction took the 2 cycles of a full-rate one | SQ_INSTS_VALU x 2 / (SIMDs x cycles) |",
    f"| VALU pipe occupancy from the instruction mix | **{(2 * (fma + add + mul) + 8 * trans + 4 * max(0.0, valu - fma - add - mul - trans)) / N_SIMD / cyc if cyc else float('nan'):.2f}** of the SIMD cycles: fp32 FMA/ADD/MUL at 2 cycles, transcendentals at 8, "
    f"everything else (conversions, compares, selects, shifts, lane swaps, 64-bit multiplies: {max(0.0, valu - fma - add - mul - trans) / max(1.0, valu) * 100:.0f} % of the instructions) at the 4 cycles measured for them "
    "(profiles/r02_valu_findings.md; the few full-rate integer forms — and/or/xor/mov/add — are priced at 4 too, so this is an upper estimate): the kernel is bound by vector issue, and half of that is half-rate integer work | SQ_INSTS_VALU_* |",
    f"| matrix pipe | busy {g('SQ_VALU_MFMA_BUSY_CYCLES') / N_SIMD / cyc if cyc else float('nan'):.3f} of SIMD cycles = {g('SQ_VALU_MFMA_BUSY_CYCLES') / 32 / wave_passes:.1f} MFMAs of 32 cycles per wavefront pass; "
    f"executed f16 matrix flops {g('SQ_VALU_MFMA_BUSY_CYCLES') / 32 * 32768 / (avg_ms * 1e-3) / 1e12:.0f} TF/s = {g('SQ_VALU_MFMA_BUSY_CYCLES') / 32 * 32768 / (avg_ms * 1e-3) / 2.5e15:.3f} of the 2.5 PF dense f16 peak; "
    + (f"a vector instruction executes during {coexec_c / max(1.0, mfma_busy_c):.2f} of the matrix cycles (co-execution {coexec:.3f} of SIMD cycles): the two pipes run side by side, "
       f"VALU or MFMA busy {valu_util + mfma_busy - coexec:.3f} of the SIMD cycles" if coexec is not None else "SQ_VALU_MFMA_COEXEC_CYCLES not collected") + " | SQ_VALU_MFMA_BUSY_CYCLES, SQ_VALU_MFMA_COEXEC_CYCLES |",
    f"| FP32 FMA / ADD / MUL / transcendental wave-instructions | {fma:.3e} / {add:.3e} / {mul:.3e} / {trans:.3e} | SQ_INSTS_VALU_* |",
    f"| EXECUTED fp32 flops (2 per FMA, 1 per ADD/MUL, x 64 lanes) as a fraction of the {PEAK_TF} TF peak | **{exec_frac:.3f}** (algorithmic flops against the same peak: {frac_trace_vec:.3f}) | counters / duration |",
    f"| SALU / LDS / SMEM / VMEM wave-instructions | {g('SQ_INSTS_SALU'):.3e} / {g('SQ_INSTS_LDS'):.3e} / {g('SQ_INSTS_SMEM'):.3e} / {g('SQ_INSTS_VMEM'):.3e} | SQ_INSTS_* |",
    f"| LDS busy | {g('SQ_LDS_IDX_ACTIVE') / 256 / cyc if cyc else float('nan'):.3f} of kernel cycles per CU; bank-conflict cycles {g('SQ_LDS_BANK_CONFLICT') / max(1.0, g('SQ_LDS_IDX_ACTIVE')):.4f} of LDS cycles | SQ_LDS_IDX_ACTIVE, SQ_LDS_BANK_CONFLICT |",
    f"| wave time split | active {g('SQ_ACTIVE_INST_ANY') / wc:.3f}, waiting on issue {g('SQ_WAIT_INST_ANY') / wc:.3f}, parked (waitcnt) {g('SQ_WAIT_ANY') / wc:.3f} of SQ_WAVE_CYCLES | SQ_ACTIVE_INST_ANY, SQ_WAIT_INST_ANY, SQ_WAIT_ANY |",
    (f"| HBM bytes per launch | fetch {fetch_b / 1e6:.2f} MB + write {write_b / 1e6:.1f} MB = {(fetch_b + write_b) / (avg_ms * 1e-3) / 1e9:.1f} GB/s ({(fetch_b + write_b) / (avg_ms * 1e-3) / 8e12 * 100:.2f} % of 8 TB/s)"
     + (f"; algorithmic bytes {roof['algorithmic_bytes_per_launch'] / 1e6:.1f} MB -> x{(fetch_b + write_b) / roof['algorithmic_bytes_per_launch']:.0f}" if roof.get("algorithmic_bytes_per_launch") else "")
     + " | FETCH_SIZE x 2 (gfx950 correction), WRITE_SIZE |") if have_traffic else "| HBM bytes per launch | not collected | |",
    f"| L2 | {g('TCC_REQ_sum'):.3e} requests, {g('TCC_ATOMIC_sum'):.3e} atomics (the fixed-point pixel accumulators), hit rate {g('TCC_HIT_sum') / max(1.0, g('TCC_HIT_sum') + g('TCC_MISS_sum')):.3f} | TCC_* |",
    "",
    "Raw per-launch counter averages: `%s_pmc.json` (launches averaged per counter: %s)." % (tag, sorted(set(launches.values()))),
]
open(os.path.join(dst, f"{tag}_summary.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
