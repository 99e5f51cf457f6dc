"""Register / LDS / scratch budget of every kernel in a built librtk*.so, read from the gfx950 code object's notes
(llvm-readelf --notes) — NOT from rocprofv3's dispatch columns, which report allocation granules of some other
dispatch on this stack ("LDS 0 B, VGPR 32" for a kernel with 24 KiB of LDS and 73 VGPRs, round 1).

    python tools/code_object_info.py [path/to/librtk.so] [--json]

Runs anywhere the ROCm LLVM tools are (no GPU needed)."""
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIELDS = ("vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "group_segment_fixed_size", "private_segment_fixed_size",
          "max_flat_workgroup_size", "uses_dynamic_stack")


def extract_code_object(so_path: str, workdir: str) -> str:
    """Unbundles the gfx950 code object of `so_path` into `workdir` (llvm-objdump writes next to its input, so the
    library is copied there first) and returns its path."""
    local = os.path.join(workdir, os.path.basename(so_path))
    shutil.copy(so_path, local)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], check=True, capture_output=True)
    for f in sorted(os.listdir(workdir)):
        if "amdgcn" in f and "gfx950" in f:
            return os.path.join(workdir, f)
    raise RuntimeError("no gfx950 code object in " + so_path)


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.splitlines()
    return dict(zip(names, out))


def kernel_info(so_path: str) -> dict:
    """{demangled kernel name: {field: value}} plus derived `waves_per_simd_by_vgpr` (512 VGPRs per SIMD lane, granule 8)."""
    with tempfile.TemporaryDirectory() as td:
        co = extract_code_object(so_path, td)
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True, check=True).stdout
    kernels, cur = [], None
    for line in notes.splitlines():
        m = re.match(r"\s+(- )?\.(\w+):\s+(.*)$", line)
        if not m:
            continue
        dash, key, val = m.groups()
        if dash and cur is not None and "name" in cur and "vgpr_count" in cur:
            kernels.append(cur)
            cur = None
        if cur is None:
            cur = {}
        if key == "name":
            cur["name"] = val.strip()
        elif key in FIELDS:
            cur[key] = val.strip()
    if cur and "name" in cur and "vgpr_count" in cur:
        kernels.append(cur)
    names = demangle([k["name"] for k in kernels])
    out = {}
    for k in kernels:
        d = {f: (int(k[f]) if k.get(f, "").lstrip("-").isdigit() else k.get(f)) for f in FIELDS if f in k}
        v = d.get("vgpr_count", 0) + d.get("agpr_count", 0)
        gran = (v + 7) // 8 * 8
        d["waves_per_simd_by_vgpr"] = min(8, 512 // gran) if gran else 8
        out[names[k["name"]]] = d
    return out


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    path = args[0] if args else os.path.join(ROOT, "raytracinginoneweekend_amd", "librtk.so")
    info = kernel_info(path)
    if "--json" in sys.argv:
        print(json.dumps(info, indent=1))
    else:
        for name, d in info.items():
            short = re.sub(r"\(.*$", "", name)
            print("%-44s VGPR %3d  SGPR %3d  spills v%d/s%d  static LDS %5d B  scratch %3d B  waves/SIMD by VGPR %d"
                  % (short, d["vgpr_count"], d["sgpr_count"], d["vgpr_spill_count"], d["sgpr_spill_count"], d["group_segment_fixed_size"],
                     d["private_segment_fixed_size"], d["waves_per_simd_by_vgpr"]))
