"""tools/asm_variant.py — build librtk variants whose DEVICE code is hand-edited assembly (for hazard hunts: one
instruction moved or one wait state inserted at an exact place, which source-level `asm volatile` cannot guarantee —
the scheduler moves independent instructions across it).

    python tools/asm_variant.py <name> [-DFLAG ...] [--edit 'KERNEL_SUBSTR:::python-regex:::replacement' ...] [--script edits.py]

Pipeline (what hipcc does, with the device half stopped at assembly):
  1. hipcc --cuda-device-only -S            -> variants/<name>.s          (then the edits are applied to it)
  2. clang -x assembler -mcpu=gfx950 -c     -> <name>.o ; lld -shared -> <name>.out ; clang-offload-bundler -> <name>.hipfb
  3. hipcc --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang <name>.hipfb -shared -> variants/librtk_<name>.so

An edit is restricted to the text of the kernels whose mangled name contains KERNEL_SUBSTR ('' = whole file) and must
match exactly `count` times (default: at least once); the tool prints every replacement it made.
An edits script is a Python file defining `edits = [(kernel_substr, regex, replacement, expected_count_or_None), ...]`.
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "raytracinginoneweekend_amd", "csrc")
OUT = os.path.join(ROOT, "raytracinginoneweekend_amd", "variants")
LLVM = "/opt/rocm/lib/llvm/bin"
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-I", os.path.join(ROOT, "include")]


def run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(" ".join(cmd) + "\n" + r.stdout + r.stderr)
        raise SystemExit(1)


def kernel_spans(text):
    """[(name, start, end)] of every function body in the assembly (label `name:` up to its `.Lfunc_end`)."""
    spans = []
    for m in re.finditer(r"^(_Z\w+):.*$", text, re.M):
        end = text.find(".Lfunc_end", m.end())
        spans.append((m.group(1), m.start(), end if end >= 0 else len(text)))
    return spans


def apply_edits(text, edits):
    for ksub, pattern, repl, count in edits:
        rx = re.compile(pattern, re.M)
        total = 0
        if ksub == "":
            text, total = rx.subn(repl, text)
        else:
            # back to front so that earlier spans keep their offsets
            for name, a, b in reversed([s for s in kernel_spans(text) if ksub in s[0]]):
                body, k = rx.subn(repl, text[a:b])
                text = text[:a] + body + text[b:]
                total += k
                if k:
                    print(f"  {k} x /{pattern}/ in {name}")
        if total == 0 or (count is not None and total != count):
            raise SystemExit(f"edit /{pattern}/ in kernels '*{ksub}*': {total} replacements, expected {count if count is not None else '>= 1'}")
    return text


def build(name, defines, edits):
    os.makedirs(OUT, exist_ok=True)
    base = os.path.join(OUT, name)
    src = os.path.join(CSRC, "rtk.hip")
    run(["/opt/rocm/bin/hipcc", *COMMON, *defines, "--cuda-device-only", "-S", src, "-o", base + ".s"])
    text = open(base + ".s").read()
    if edits:
        text = apply_edits(text, edits)
        open(base + ".s", "w").write(text)
    run([os.path.join(LLVM, "clang"), "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", base + ".s", "-o", base + ".o"])
    run([os.path.join(LLVM, "lld"), "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-o", base + ".out", base + ".o"])
    run([os.path.join(LLVM, "clang-offload-bundler"), "-type=o", "-bundle-align=4096", "-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950",
         "-input=/dev/null", "-input=" + base + ".out", "-output=" + base + ".hipfb"])
    lib = os.path.join(OUT, f"librtk_{name}.so")
    run(["/opt/rocm/bin/hipcc", *COMMON, *defines, "--cuda-host-only", "-Xclang", "-fcuda-include-gpubinary", "-Xclang", base + ".hipfb", "-shared", src, "-o", lib])
    for ext in (".o", ".out", ".hipfb"):
        os.remove(base + ext)
    print("built", lib)
    return lib


def main(argv):
    name, defines, edits = argv[0], [], []
    i = 1
    while i < len(argv):
        a = argv[i]
        if a.startswith("-D"):
            defines.append(a)
        elif a == "--edit":
            i += 1
            k, p, r = argv[i].split(":::")
            edits.append((k, p, r.encode().decode("unicode_escape"), None))
        elif a == "--script":
            i += 1
            ns = {}
            exec(open(argv[i]).read(), ns)
            edits.extend(ns["edits"])
        else:
            raise SystemExit("unknown argument " + a)
        i += 1
    build(name, defines, edits)


if __name__ == "__main__":
    main(sys.argv[1:])
