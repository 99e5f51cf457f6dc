#!/usr/bin/env python3
"""tools/block_budget.py [--ms MS --passes P] [--out FILE] — a per-block VALU budget of one pass of the render kernel.

Static side: the kernel is compiled exactly as build.py does plus -gline-tables-only (debug line tables do not change
the generated code; the instruction count is checked against the shipped librtk.so), disassembled, and every instruction
of render_kernel<LDS list> is attributed to a block of the pass through its INLINE CHAIN (llvm-symbolizer --inlines):
filter tests, word-loop control, exact root test, candidate pop, Philox, primary ray, scatter per material, ...
Issue cost per instruction class: the wave64 issue cycles measured on the GPU by tools/valu_cycles.hip
(profiles/r02_valu_cycles_*.txt): 2 for the full-rate fp32/integer ops, 8 for transcendentals, 4 for everything else.

Dynamic side: how often a wave executes each block per pass, counted by the 64-lane CPU emulation of the same source
(tests/cpu_emul, RTK_STAT counters) on the stock scene.

Result: static VALU x issue cycles x executions per pass, per block, and the sum next to the measured cycles per pass
(kernel time x clock x SIMDs / passes) when --ms/--passes are given.  CPU only.
"""
from __future__ import annotations

import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "raytracinginoneweekend_amd", "csrc")
LLVM = "/opt/rocm/lib/llvm/bin"
KERNEL = "_Z13render_kernelILb1ELb0ELi4EEvN3rtk10RenderArgsE"

FAST = {"v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_and_b32", "v_or_b32", "v_xor_b32",
        "v_lshrrev_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_mov_b32", "v_fmaak_f32", "v_fmamk_f32", "v_mov_b64",
        "v_madak_f32", "v_madmk_f32", "v_mac_f32"}
TRANS = {"v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32", "v_rcp_iflag_f32"}


def issue_cycles(op: str) -> int:
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    return 2 if base in FAST else 8 if base in TRANS else 4


def anchor(path: str, text: str) -> int:
    for i, ln in enumerate(open(path), 1):
        if text in ln:
            return i
    raise SystemExit(f"anchor {text!r} not found in {path}")


def build_code_object(tmp: str) -> str:
    from raytracinginoneweekend_amd import build as B
    obj = os.path.join(tmp, "rtk_dev.o")
    flags = [f for f in B.HIP_FLAGS if f != "-Wall"]      # includes -mllvm -amdgpu-mfma-vgpr-form=1
    subprocess.run([B.HIPCC, *flags, "--cuda-device-only", "-gline-tables-only", "-c", "-I", os.path.join(ROOT, "include"),
                    os.path.join(CSRC, "rtk.hip"), "-o", obj], check=True)
    co = os.path.join(tmp, "rtk.co")
    subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={obj}",
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
    return co


def disassemble(co: str):
    """[(address, opcode)] of the kernel."""
    txt = subprocess.run([f"{LLVM}/llvm-objdump", "-d", f"--disassemble-symbols={KERNEL}", co], check=True, capture_output=True, text=True).stdout
    out = []
    for ln in txt.splitlines():
        m = re.match(r"\s+(\S+).*//\s*([0-9A-Fa-f]{12}):", ln)
        if m:
            out.append((int(m.group(2), 16), m.group(1)))
    return out


def shipped_instruction_count() -> int | None:
    """Instruction count of the same kernel in the shipped librtk.so (None if it cannot be extracted)."""
    lib = os.path.join(ROOT, "raytracinginoneweekend_amd", "librtk.so")
    if not os.path.exists(lib):
        return None
    with tempfile.TemporaryDirectory() as t:
        co = os.path.join(t, "lib.co")
        r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={lib}",
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], capture_output=True)
        if r.returncode != 0 or not os.path.exists(co):
            # a linked library keeps its bundle in .hip_fatbin: take the section out first
            fat = os.path.join(t, "fat.bin")
            r = subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat], capture_output=True)
            if r.returncode != 0:
                return None
            r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                                "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], capture_output=True)
            if r.returncode != 0:
                return None
        return len(disassemble(co))


def symbolize(co: str, addrs):
    """address -> [(function, file, line)] innermost first."""
    inp = "\n".join(hex(a) for a in addrs) + "\n"
    txt = subprocess.run([f"{LLVM}/llvm-symbolizer", f"--obj={co}", "--inlines", "--functions=short"], input=inp, check=True,
                         capture_output=True, text=True).stdout
    chains, cur = [], []
    lines = txt.split("\n")
    i = 0
    while i < len(lines):
        if lines[i] == "":
            if cur:
                chains.append(cur)
                cur = []
            i += 1
            continue
        fn = lines[i]
        loc = lines[i + 1] if i + 1 < len(lines) else "?:0:0"
        m = re.match(r"(.*):(\d+):(\d+)$", loc)
        cur.append((fn, os.path.basename(m.group(1)) if m else "?", int(m.group(2)) if m else 0))
        i += 2
    if cur:
        chains.append(cur)
    if len(chains) != len(addrs):
        raise SystemExit(f"symbolizer returned {len(chains)} chains for {len(addrs)} addresses")
    return dict(zip(addrs, chains))


def make_classifier():
    R = os.path.join(CSRC, "rtk_render.hpp")
    I = os.path.join(CSRC, "rtk_intersect.hpp")
    D = os.path.join(CSRC, "rtk_device.hpp")
    r_refill = anchor(R, "// ---- lanes without a live path")
    r_assign = anchor(R, "// ---- hand out units")
    r_start = anchor(R, "// ---- primary rays")
    r_loopctl = anchor(R, "if (!wave_any(alive)) {")
    r_seg = anchor(R, "// ---- one segment")
    r_alive = anchor(R, "if (alive) {")
    r_sky = anchor(R, "if (idx < 0) {")
    r_hit = anchor(R, "const uint32_t bounce = static_cast<uint32_t>(A.max_depth - depth_left) + 1u;")
    r_end = anchor(R, "atomicAdd(A.stats + 0")
    i_calls = anchor(I, "RTK_STAT(lane, closest_hit_calls)")
    i_words = anchor(I, "auto scan_words = [&]()")
    i_classloop = anchor(I, "if (!has_general) {")
    i_mixedloop = anchor(I, "f16x8 bc[2], bg0[2], bg1[2];")
    i_tail = anchor(I, "if (short_tail) {          // before the word loops")
    i_bfrag = anchor(I, "// B fragments: lane l supplies")
    i_unsafe = anchor(I, "// lanes whose ray the filter cannot judge")
    i_exact = anchor(I, "// exact stage: every lane pops")
    d_lambert = anchor(D, "if (type == 1) {")
    d_metal = anchor(D, "if (type == 2) {")
    d_glass = anchor(D, "if (type == 3) {")
    d_none = anchor(D, "// Type::None")
    def classify(chain):
        fns = [f for f, _, _ in chain]
        rw = next(((f, fl, l) for f, fl, l in chain if f.startswith("render_wave")), None)
        if rw is None:
            return "kernel prologue / epilogue (LDS staging, once per wave)"
        line = rw[2]
        if line >= r_end:
            return "kernel prologue / epilogue (LDS staging, once per wave)"
        if "closest_hit" in fns:
            if "sphere_root_exact" in fns:
                return "exact: root test (Sphere3.hpp:31-49)"
            lam = [l for f, fl, l in chain if f == "operator()" and fl == "rtk_intersect.hpp"]
            l = max(lam) if lam else 0          # the scan_words frame (the `keep` lambda sits above it in the file)
            if lam:
                if l >= i_exact:
                    return "exact: candidate pop (mask word -> index, loop control)"
                if l >= i_unsafe:
                    return "filter: unsafe-ray override"
                if l >= i_mixedloop:
                    return "filter: word loop of scenes with general words (not entered on the stock scene)"
                if l >= i_classloop:
                    return "filter: class word loop (A fragment, 2 MFMA, 2 FMA per test, mask assembly and store)"
                if l >= i_bfrag:
                    return "filter: per-pass ray coefficients (f16 splits, lane exchange)"
                if l >= i_tail:
                    return "filter: short last word on the vector ALUs (11 FMA per test)"
                return "filter: per-pass ray coefficients (f16 splits, lane exchange)"
            return "filter: per-ray setup (unit direction, safety)"
        if line < r_assign:
            return "refill: next sample of the run / flush the run's sums (3 x 64-bit atomic)"
        if line < r_start:
            return "assign: hand out packet units, draw packets"
        if line < r_loopctl:
            if "philox4x32_10" in fns or "u01" in fns:
                return "start: Philox4x32-10 + uniforms"
            return "start: primary ray (camera from LDS, disk sample)"
        if line < r_seg:
            return "loop control"
        if line < r_alive:
            return "closest_hit: other"
        if line < r_sky:
            return "segment: unit_vector(d)"
        if line < r_hit:
            return "sky: colour, 32.32 fixed-point sums"
        if "philox4x32_10" in fns or "u01" in fns:
            return "hit: Philox4x32-10 + uniforms"
        if "hit_and_scatter" in fns:
            hs = [l for f, fl, l in chain if f == "hit_and_scatter" and fl == "rtk_device.hpp"]
            l = hs[0] if hs else 0
            if l >= d_none:
                return "hit: record (p, normal, front face) + Type::None"
            if l >= d_glass:
                return "hit: scatter glass"
            if l >= d_metal:
                return "hit: scatter metal"
            if l >= d_lambert:
                return "hit: scatter lambertian"
            return "hit: record (p, normal, front face) + Type::None"
        return "hit: operand loads, throughput product, bookkeeping"

    return classify


def emulation_frequencies(width=96, height=64, spp=8):
    """Executions of each block per pass of a wave, from the 64-lane emulation on the stock scene."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import raytracinginoneweekend_amd as rt
    import emul
    emul.lib(64).emul_force_staged(0)       # libemul64.so defaults to the staged large-scene path; this budget is for the LDS path
    try:
        e = emul.Emul(rt.stock_scene(), lanes=64)
        emul.wave_stats(64)
        _, st = e.render(rt.stock_camera(width / height), width, height, spp, 50, seed=1, threads=4)
        s = emul.wave_stats(64)
    finally:
        emul.lib(64).emul_force_staged(1)
    p = float(s["closest_hit_calls"])
    f = {
        "filter: class word loop (A fragment, 2 MFMA, 2 FMA per test, mask assembly and store)": s["filter_words"] / p,
        "filter: word loop of scenes with general words (not entered on the stock scene)": 0.0,
        "filter: short last word on the vector ALUs (11 FMA per test)": 1.0,
        "filter: per-pass ray coefficients (f16 splits, lane exchange)": 1.0,
        "filter: word loop (descriptor, per-word constants, mask assembly and store)": s["filter_words"] / p,
        "filter: general tests of the partial last word": 1.0,
        "filter: per-ray setup (unit direction, safety)": 1.0,
        "filter: unsafe-ray override": 0.0,
        "exact: root test (Sphere3.hpp:31-49)": s["flush_iters"] / p,
        "exact: candidate pop (mask word -> index, loop control)": s["flush_iters"] / p,
        "closest_hit: other": 1.0,
        "refill: next sample of the run / flush the run's sums (3 x 64-bit atomic)": s["pass_refill_block"] / p,
        "assign: hand out packet units, draw packets": s["pass_assign_iters"] / p,
        "start: Philox4x32-10 + uniforms": s["pass_start_block"] / p,
        "start: primary ray (camera from LDS, disk sample)": s["pass_start_block"] / p,
        "loop control": 1.0,
        "segment: unit_vector(d)": 1.0,
        "sky: colour, 32.32 fixed-point sums": s["pass_sky_block"] / p,
        "hit: Philox4x32-10 + uniforms": s["pass_hit_block"] / p,
        "hit: operand loads, throughput product, bookkeeping": s["pass_hit_block"] / p,
        "hit: record (p, normal, front face) + Type::None": s["pass_hit_block"] / p,
        "hit: scatter lambertian": s["pass_lambert"] / p,
        "hit: scatter metal": s["pass_metal"] / p,
        "hit: scatter glass": s["pass_glass"] / p,
        "kernel prologue / epilogue (LDS staging, once per wave)": 0.0,
    }
    meta = {"frame": f"{width}x{height}x{spp}", "passes": int(p), "rays_per_pass": st["rays"] / p, "exact_per_ray": st["exact_tests"] / max(1, st["rays"]) if "exact_tests" in st else None}
    return f, meta


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ms", type=float, default=None, help="measured kernel time of one launch")
    ap.add_argument("--passes", type=float, default=None, help="wave passes of that launch (rtk_stats.wave_passes)")
    ap.add_argument("--clock-ghz", type=float, default=2.4)
    ap.add_argument("--simds", type=int, default=1024)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()

    with tempfile.TemporaryDirectory() as tmp:
        co = build_code_object(tmp)
        ins = disassemble(co)
        chains = symbolize(co, [ad for ad, _ in ins])
    classify = make_classifier()
    blocks = collections.OrderedDict()
    for ad, op in ins:
        b = blocks.setdefault(classify(chains[ad]), {"valu": 0, "cyc": 0, "salu": 0, "lds": 0, "vmem": 0, "slow": collections.Counter()})
        if op.startswith("v_mfma"):
            b["mfma"] = b.get("mfma", 0) + 1
            # v_mfma_f32_32x32x16_f16 occupies the matrix pipe for 32 cycles; vector instructions of the SIMD's other waves run
            # beside it (SQ_VALU_MFMA_COEXEC_CYCLES: 0.67 of the matrix cycles), so this column is an upper bound for the loop
            b["cyc"] += 32
        elif op.startswith("v_"):
            c = issue_cycles(op)
            b["valu"] += 1
            b["cyc"] += c
            if c != 2:
                b["slow"][re.sub(r"_(e32|e64)$", "", op)] += 1
        elif op.startswith("ds_"):
            b["lds"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            b["vmem"] += 1
        elif op.startswith("s_") and not op.startswith(("s_waitcnt", "s_nop")):
            b["salu"] += 1
    freq, meta = emulation_frequencies()
    shipped = shipped_instruction_count()
    # the class word loop is written five words per trip plus a remainder loop: its block holds several copies of the body
    # (two MFMAs each); the emulation counts WORDS per pass, so the static figures are brought back to one word
    for name, b in blocks.items():
        if name.startswith("filter: class word loop") and b.get("mfma", 0) > 2:
            copies = b["mfma"] // 2
            b["copies"] = copies
            for k in ("valu", "cyc", "salu", "lds", "vmem"):
                b[k] = b[k] / copies
            b["slow"] = collections.Counter({k: round(v / copies, 1) for k, v in b["slow"].items()})

    L = []
    L.append("# Per-block VALU budget of one pass of the render kernel (LDS list path)\n")
    L.append("Generated by `python tools/block_budget.py` (CPU only).  Static counts: the kernel compiled with build.py's flags, every instruction "
             "attributed to a block through its inline chain (`llvm-symbolizer --inlines`).  Issue cycles per wave64 instruction as measured by "
             "`tools/valu_cycles.hip` (2 full-rate fp32/int, 8 transcendental, 4 the rest).  Executions per pass: the 64-lane CPU emulation of the "
             f"same source on the stock scene ({meta['frame']}, {meta['passes']} passes, {meta['rays_per_pass']:.1f} live rays per pass of 64 lanes).  "
             "A block that any lane enters costs the wave the whole block, so the frequencies are per wave, not per lane.\n")
    L.append(f"Instructions in the kernel: {len(ins)}" + (f" (shipped librtk.so: {shipped})" if shipped is not None else "") + ".\n")
    L.append("| block | VALU | issue cycles (static) | SALU | LDS | VMEM | executions / pass | cycles / pass | share | instructions above 2 cycles |")
    L.append("|---|---:|---:|---:|---:|---:|---:|---:|---:|---|")
    total = sum(b["cyc"] * freq.get(k, 1.0) for k, b in blocks.items())
    rows = sorted(blocks.items(), key=lambda kv: -kv[1]["cyc"] * freq.get(kv[0], 1.0))
    for k, b in rows:
        fr = freq.get(k, 1.0)
        dyn = b["cyc"] * fr
        slow = " ".join(f"{o}:{n}" for o, n in b["slow"].most_common(6))
        if b.get("mfma"):
            slow = f"v_mfma_f32_32x32x16_f16:{b['mfma']} (32 cycles each) " + slow
        note = f" (one of the {b['copies']} copies of the body: five words per trip + remainder loop)" if b.get("copies") else ""
        L.append(f"| {k}{note} | {b['valu']:.0f} | {b['cyc']:.0f} | {b['salu']:.0f} | {b['lds']:.0f} | {b['vmem']:.0f} | {fr:.2f} | {dyn:.0f} | {100 * dyn / total:.1f} % | {slow} |")
    L.append(f"| **sum** | {sum(b['valu'] for b in blocks.values()):.0f} | | | | | | **{total:.0f}** | 100 % | |")
    L.append("")
    if a.ms and a.passes:
        cyc = a.ms * 1e-3 * a.clock_ghz * 1e9 * a.simds / a.passes
        L.append(f"Measured: {a.ms} ms per launch, {a.passes:.4g} wave passes -> {cyc:.0f} SIMD cycles per pass at {a.clock_ghz} GHz over {a.simds} SIMDs; "
                 f"the static estimate explains {100 * total / cyc:.0f} % of it (the rest: SALU/branch issue that does not overlap, LDS waits of the "
                 "exact stage, drain at the end of the launch, and clock below nominal).\n")
    txt = "\n".join(L) + "\n"
    if a.out:
        open(a.out, "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main()
