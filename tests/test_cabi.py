"""The C-ABI library loads and exports every symbol include/rtk.h declares; argument checking and the
no-device error path work without a GPU.  No compute calls here."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT
import raytracinginoneweekend_amd as rt


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "rtk.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rtk_[a-z0-9_]+)\s*\(", text)))


def test_header_and_python_agree():
    assert declared_symbols() == sorted(rt.C_ABI_SYMBOLS)


def test_every_declared_symbol_is_exported():
    lib = C.CDLL(rt.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(lib, name), name


def test_struct_sizes_match_header():
    assert C.sizeof(rt.rtk_camera) == 88
    assert C.sizeof(rt.rtk_scene) == 8 + 11 * 8
    assert C.sizeof(rt.rtk_params) == 56
    assert C.sizeof(rt.rtk_stats) == 80


def test_local_tile_count():
    l = rt.lib()
    # 1200 x 800 in 8 x 8 tiles = 150 x 100 = 15000 tiles
    assert l.rtk_local_tile_count(1200, 800, 8, 8, 0, 1) == 15000
    assert sum(l.rtk_local_tile_count(1200, 800, 8, 8, r, 8) for r in range(8)) == 15000
    # ragged: 13 x 9 -> 2 x 2 tiles
    assert [l.rtk_local_tile_count(13, 9, 8, 8, r, 3) for r in range(3)] == [2, 1, 1]
    assert l.rtk_local_tile_count(13, 9, 0, 0, 0, 1) == 4          # 0,0 -> 8 x 8
    assert l.rtk_local_tile_count(13, 9, 8, 8, 3, 3) == -1         # rank out of range
    assert l.rtk_local_tile_count(0, 9, 8, 8, 0, 1) == -1


def test_null_arguments_are_rejected_without_a_device():
    l = rt.lib()
    assert l.rtk_context_create(0, None) == -1
    assert b"null" in l.rtk_last_error()
    assert l.rtk_set_scene(None, None) == -1
    assert l.rtk_synchronize(None) == -1


@pytest.mark.skipif(rt.lib().rtk_device_count() > 0, reason="a GPU is present")
def test_no_device_fails_loudly():
    with pytest.raises(rt.RtkError, match="no CPU fallback"):
        rt.Context(0)


def test_run_schedule_partitions_the_samples():
    """rtk_run_schedule (host only): runs tile [0, spp) exactly and shrink towards the end; finer for fewer pixels."""
    import numpy as np
    for spp, W, H, rl in [(500, 1200, 800, 0), (1000, 3840, 2160, 0), (10, 400, 225, 0), (1, 16, 16, 0), (500, 120, 80, 0), (500, 1200, 800, 8), (7, 64, 48, 3), (40, 64, 48, 40), (5, 64, 48, 9)]:
        r = rt.run_schedule(spp, W, H, rl)
        assert r[0, 0] == 0 and (r[:, 1] >= 1).all()
        assert np.array_equal(r[1:, 0], np.cumsum(r[:, 1])[:-1]) and r[:, 1].sum() == spp
        if rl == 0:
            assert (np.diff(r[:, 1].astype(int)) <= 0).all()          # decreasing
            assert r[-1, 1] == 1 or spp < 8                            # the frame ends on single-sample units
        else:
            assert (r[:-1, 1] == min(rl, spp)).all()
    # C2 on one GPU: k = 32 -> first run 16 samples, 106 runs, the last 31 samples one by one; an eighth of it (one of 8
    # ranks): k = 250 -> single samples almost from the start; C3: k = 32 -> first run 32 samples
    c2 = rt.run_schedule(500, 1200, 800)
    assert c2[0, 1] == 16 and len(c2) == 106 and (c2[:, 1] == 1).sum() == 31
    c2_8 = rt.run_schedule(500, 1200, 100)
    assert c2_8[0, 1] == 2 and (c2_8[:, 1] == 1).sum() == 250
    assert rt.run_schedule(1000, 3840, 2160)[0, 1] == 32
    # run_len < 0: decreasing runs with k = -run_len
    assert rt.run_schedule(500, 1200, 800, -64)[0, 1] == 8 and len(rt.run_schedule(500, 1200, 800, -64)) == 172
    assert rt.lib().rtk_run_schedule(0, 10, 10, 0, None, 0) == -1


def test_built_library_is_not_older_than_its_sources():
    """The GPU box runs the in-tree librtk.so as it is (no build there): it must be newer than every kernel source."""
    csrc = os.path.join(ROOT, "raytracinginoneweekend_amd", "csrc")
    srcs = [os.path.join(csrc, f) for f in os.listdir(csrc)] + [os.path.join(ROOT, "include", "rtk.h")]
    lib_time = os.path.getmtime(rt.LIB_PATH)
    stale = [os.path.basename(s) for s in srcs if os.path.getmtime(s) > lib_time]
    assert not stale, "rebuild librtk.so (python -m raytracinginoneweekend_amd.build): newer sources %s" % stale


def test_rccl_library_offers_what_rtk_multi_opens_on_demand():
    """rtk_multi_create (distinct devices) dlopens librccl.so.1 and resolves these symbols; nothing is linked."""
    import subprocess
    needed = subprocess.run(["readelf", "-d", rt.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "rccl" not in needed
    l = C.CDLL("librccl.so.1")
    for name in ("ncclCommInitAll", "ncclCommDestroy", "ncclGroupStart", "ncclGroupEnd", "ncclSend", "ncclRecv", "ncclGetErrorString"):
        assert hasattr(l, name), name


def _disassembly(tmp_path, lib=None):
    """Disassembly of the gfx950 code object inside the built library (skips where the ROCm LLVM tools are missing)."""
    import subprocess
    llvm = "/opt/rocm/lib/llvm/bin"
    lib = lib or os.path.join(ROOT, "raytracinginoneweekend_amd", "librtk.so")
    if not (os.path.exists(lib) and os.path.exists(f"{llvm}/llvm-objdump")):
        pytest.skip("needs the built library and the ROCm LLVM tools")
    fat, co = str(tmp_path / "fat.bin"), str(tmp_path / "rtk.co")
    subprocess.run([f"{llvm}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
    subprocess.run([f"{llvm}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
    return subprocess.run([f"{llvm}/llvm-objdump", "-d", co], check=True, capture_output=True, text=True).stdout


def _mfma_followers(txt):
    """[(mfma line, mnemonic and operands of the instruction after it)] for every v_mfma in a disassembly."""
    import re
    ins = [m.group(1).strip() for m in (re.match(r"\s+([a-z_0-9]+\s[^/]*|[a-z_0-9]+)\s*//", ln) for ln in txt.splitlines()) if m]
    return [(a, b) for a, b in zip(ins, ins[1:] + [""]) if a.startswith("v_mfma")]


def test_matrix_instructions_are_issued_back_to_back(tmp_path):
    """profiles/r03_hazard_notes.md: two MFMAs of a wave that name the same A registers with a scalar-ALU instruction (or a
    waiting s_waitcnt) between them gave wrong candidate words in the render kernel (found by moving single instructions
    of the built code object: tools/asm_variant.py).  The filter's MFMAs are asm statements now, so the compiler cannot
    schedule anything between them; this reads the SHIPPED code object: every v_mfma is directly followed by another
    v_mfma or by the statement's own s_nop 11 (the wait states an 8-pass result needs)."""
    pairs = _mfma_followers(_disassembly(tmp_path))
    assert len(pairs) >= 12          # render and intersect kernel: class-only loop (2), mixed loop (2 + 4)
    for mfma, nxt in pairs:
        assert nxt.startswith("v_mfma") or nxt.replace(" ", "") == "s_nop11", (mfma, nxt)


def test_the_known_bad_build_violates_the_back_to_back_rule(tmp_path):
    """The lint above must be able to fail: the experiment build with a scalar move between the MFMAs (the form
    tests/test_gpu_render.py::test_known_bad_loop_form_is_caught renders with) is caught by it."""
    bad = os.path.join(ROOT, "raytracinginoneweekend_amd", "variants", "librtk_known_bad.so")
    if not os.path.exists(bad):
        pytest.skip("variants/librtk_known_bad.so not built (__graft_entry__.build() makes it)")
    pairs = _mfma_followers(_disassembly(tmp_path, bad))
    assert any(nxt.startswith("s_mov_b32") for _, nxt in pairs)


def test_no_mfma_result_overlaps_its_operands(tmp_path):
    """A 32x32 MFMA's result registers must not overlap its A or B operand, and its C operand only exactly (the matrix
    pipe reads the operands over several passes).  The result tiles are early-clobber outputs of the asm statements in
    rtk_intersect.hpp (mfma_pair, mfma_quad); this test reads the shipped code object to see that it holds."""
    import re
    txt = _disassembly(tmp_path)

    def regs(op):
        m = re.fullmatch(r"([av])\[(\d+):(\d+)\]", op)
        if m:
            return m.group(1), set(range(int(m.group(2)), int(m.group(3)) + 1))
        m = re.fullmatch(r"([av])(\d+)", op)
        return (m.group(1), {int(m.group(2))}) if m else (None, set())

    n = 0
    for ln in txt.splitlines():
        m = re.match(r"\s+(v_mfma_\S+)\s+([^/]+)", ln)
        if not m:
            continue
        ops = [o.strip() for o in m.group(2).split(",")][:4]
        (df, d), (af, a), (bf, b), (cf, c) = (regs(o) for o in ops)
        n += 1
        assert not (df == af and d & a) and not (df == bf and d & b), ln
        assert not (df == cf and d & c) or d == c, ln
    assert n >= 8          # the render and the intersect kernel: class loop and mixed loop


def test_render_kernels_use_no_scratch_memory():
    """No kernel of the library spills vector registers to scratch memory (the LDS render kernel sits exactly at the 128
    registers that four waves per SIMD allow; a change that pushes it over shows up here, not as a slower kernel).  SGPR
    spills to VGPR lanes are fine."""
    import json, subprocess, sys
    lib = os.path.join(ROOT, "raytracinginoneweekend_amd", "librtk.so")
    if not (os.path.exists(lib) and os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf")):
        pytest.skip("needs the built library and the ROCm LLVM tools")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "code_object_info.py"), lib, "--json"], check=True, capture_output=True, text=True).stdout
    info = json.loads(out)
    kernels = info if isinstance(info, list) else info.get("kernels", info)
    seen = 0
    for k in (kernels.values() if isinstance(kernels, dict) else kernels):
        seen += 1
        assert k["vgpr_spill_count"] == 0 and k["private_segment_fixed_size"] == 0, k
    assert seen >= 10
