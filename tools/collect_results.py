"""Turn what tools/gpu_session.sh left under gpurun_out/ into one markdown summary, profiles/<tag>_results.md
(the tracked place the numbers in DESIGN.md are quoted from).
usage: python tools/collect_results.py [tag]"""
import glob, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
src = os.path.join(ROOT, "gpurun_out")
out = [f"# GPU results, tag `{tag}` (collected from gpurun_out/ by tools/collect_results.py)", ""]


def bench_lines():
    rows = []
    for path in sorted(glob.glob(os.path.join(src, f"{tag}_*.json"))):
        for line in open(path, errors="replace"):
            if line.startswith("{") and '"metric"' in line:
                try:
                    rows.append((os.path.basename(path), json.loads(line)))
                except json.JSONDecodeError:
                    pass
    return rows


rows = bench_lines()
if rows:
    out += ["## bench.py lines", "", "| file | workload | GPUs | Mrays/s | ms/step | roofline frac | kernel ms | lane util | exact/ray | frames in flight |", "|---|---|---|---|---|---|---|---|---|---|"]
    for name, j in rows:
        r = j.get("roofline", {})
        fmt = lambda v, f: (f % v) if isinstance(v, (int, float)) else "—"
        out.append("| %s | %s | %d | %.0f | %.2f | %s | %s | %s | %s | %s |" % (
            name, j["metric"].replace("Mrays/s (whole node), ", "")[:70], j["n_gpus"], j["value"], j["ms_per_step"], fmt(r.get("frac"), "%.3f"),
            fmt(r.get("kernel_ms"), "%.2f"), fmt(r.get("lane_utilisation"), "%.3f"), fmt(r.get("exact_tests_per_ray"), "%.2f"),
            j.get("config", {}).get("frames_in_flight", 1)))
        if "cpu_baseline" in j:
            c = j["cpu_baseline"]
            out.append(f"|  | cpu_baseline ({c['kind']}, {c['cores']} core): {c['value']:.3f} {c['unit']} — {c['sample']} | | | | | | | | |")
    out.append("")
for name, title in ((f"{tag}_ab.txt", "A/B of kernel variants (tools/ab_bench.py)"), (f"{tag}_valu_rates.txt", "Instruction issue rates, 4 waves per SIMD (tools/valu_rates.hip)"),
                    (f"{tag}_valu_rates_1wave.txt", "Instruction issue rates, 1 wave per SIMD")):
    p = os.path.join(src, name)
    if os.path.exists(p):
        out += [f"## {title}", "", "```", open(p, errors="replace").read().strip(), "```", ""]
for p in sorted(glob.glob(os.path.join(src, f"{tag}_tests*.log"))):
    tail = [l for l in open(p, errors="replace").read().strip().splitlines() if re.search(r"passed|failed|error", l)]
    out += [f"## {os.path.basename(p)}", "", "```", *(tail[-3:] or ["(no pytest summary line)"]), "```", ""]
dst = os.path.join(ROOT, "profiles", f"{tag}_results.md")
open(dst, "w").write("\n".join(out) + "\n")
print("wrote", dst, "from", len(rows), "bench lines")
