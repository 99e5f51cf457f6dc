"""ctypes view of tests/cpu_emul/libemul.so: the librtk DEVICE code compiled for the CPU (one lane per wave).
Test infrastructure; see tests/cpu_emul/rtk_host_shim.hpp."""
import ctypes as C
import os
import subprocess

import numpy as np

import raytracinginoneweekend_amd as rt

HERE = os.path.dirname(os.path.abspath(__file__))
_libs = {}

STAT_FIELDS = ("closest_hit_calls", "filter_words", "compact_iters", "early_flushes", "flush_iters", "pass_refill_block",
               "pass_assign_iters", "pass_start_block", "pass_sky_block", "pass_hit_block", "pass_lambert", "pass_metal",
               "pass_glass", "pass_unit_flush_block", "packet_draws", "class_words", "tail_tests", "clamp_fractions")


def lib(lanes=1):
    """lanes = 1: one lane per wave (fast).  lanes = 64: 64 fibers per wave meeting at the votes (cross-lane logic)."""
    if lanes not in _libs:
        subprocess.run(["make", "-C", os.path.join(HERE, "cpu_emul")], check=True, capture_output=True)
        l = C.CDLL(os.path.join(HERE, "cpu_emul", {1: "libemul.so", 64: "libemul64.so"}[lanes]))
        vp = C.c_void_p
        l.emul_create.restype = vp
        l.emul_create.argtypes = [C.POINTER(rt.rtk_scene)]
        l.emul_destroy.argtypes = [vp]
        l.emul_class_words.argtypes = [vp]
        l.emul_local_words.argtypes = [vp]
        l.emul_local_frames.argtypes = [C.c_int]
        l.emul_scene_info.argtypes = [vp, vp, vp]
        l.emul_intersect.argtypes = [vp, vp, C.c_uint64, C.c_float, C.c_float, vp, vp, vp, vp, vp, C.c_int, C.c_int]
        l.emul_grid_info.argtypes = [vp, vp]
        l.emul_scatter.argtypes = [vp, vp, vp, vp, vp, C.c_uint64, vp]
        l.emul_primary.argtypes = [vp, C.c_int, C.c_int, C.c_uint64, vp, C.c_uint64, vp]
        l.emul_philox.argtypes = [C.c_uint64, vp, C.c_uint64, vp]
        l.emul_render_tiles.argtypes = [vp, vp, C.POINTER(rt.rtk_params), vp, vp, C.c_int, vp, C.c_int]
        l.emul_get_stats.argtypes = [vp, C.c_int]
        _libs[lanes] = l
    return _libs[lanes]


def wave_stats(lanes=1, reset=True):
    """Wave-level block execution counts of the emulated kernels since the last reset."""
    out = np.zeros(len(STAT_FIELDS), np.uint64)
    got = lib(lanes).emul_get_stats(out.ctypes.data, len(out))
    assert got == lanes
    if reset:
        lib(lanes).emul_reset_stats()
    return dict(zip(STAT_FIELDS, (int(x) for x in out)))


class Emul:
    def __init__(self, scene: dict, lanes: int = 1):
        self._l = lib(lanes)
        self.lanes = lanes
        arrs = {k: np.ascontiguousarray(scene[k], np.float32) for k in rt.SCENE_FIELDS[:10]}
        arrs["mat_type"] = np.ascontiguousarray(scene["mat_type"], np.int32)
        s = rt.rtk_scene()
        s.n = len(arrs["cx"])
        for k in rt.SCENE_FIELDS[:10]:
            setattr(s, k, arrs[k].ctypes.data_as(C.POINTER(C.c_float)))
        s.mat_type = arrs["mat_type"].ctypes.data_as(C.POINTER(C.c_int32))
        self.n = s.n
        self._h = C.c_void_p(self._l.emul_create(C.byref(s)))

    def __del__(self):
        try:
            self._l.emul_destroy(self._h)
        except Exception:
            pass

    def scene_info(self):
        b = np.zeros(8, np.float32)
        f = np.zeros(self.n, np.uint8)
        self._l.emul_scene_info(self._h, b.ctypes.data, f.ctypes.data)
        return {"c_max": float(b[0]), "r_max2": float(b[1]), "r_min": float(b[2]), "sigma": float(b[3]), "cls_cy": float(b[4]),
                "cls_r2": float(b[5]), "acc_eps": float(b[6]), "kinds": int(b[7])}, f

    def class_words(self) -> int:
        """Words of 32 spheres that got a 16-term class tile (shared cy and radius); 0 on the staged path."""
        return int(self._l.emul_class_words(self._h))

    def local_words(self) -> int:
        """Words with a tile in their own frame, when the scene streams through the matrix-pipe filter (else 0)."""
        return int(self._l.emul_local_words(self._h))

    def grid_info(self):
        a = np.zeros(9, np.int64)
        self._l.emul_grid_info(self._h, a.ctypes.data)
        return {"usable": bool(a[0]), "dims": (int(a[1]), int(a[2]), int(a[3])), "n_always": int(a[4]), "items": int(a[5]),
                "items_per_cell": a[6] / 1000.0, "cells_per_sphere": a[7] / 1000.0, "zones": int(a[8])}

    def intersect(self, rays, t_min=0.001, t_max=float("inf"), want_flags=False, ulp_noise=0, accel=0):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        n = rays.shape[0]
        hit = np.zeros(n, np.uint8); idx = np.zeros(n, np.int32); t = np.zeros(n, np.float32)
        flags = np.zeros((n, self.n), np.uint8) if want_flags else None
        ne = C.c_uint64(0)
        self._l.emul_intersect(self._h, rays.ctypes.data, n, t_min, t_max, hit.ctypes.data, idx.ctypes.data, t.ctypes.data,
                               flags.ctypes.data if want_flags else None, C.addressof(ne), ulp_noise, accel)
        self.last_exact = ne.value
        return (hit, idx, t, flags) if want_flags else (hit, idx, t)

    def scatter(self, rays, index, t, u4):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        n = rays.shape[0]
        index = np.ascontiguousarray(index, np.int32); t = np.ascontiguousarray(t, np.float32)
        u4 = np.ascontiguousarray(u4, np.float32).reshape(n, 4)
        out = np.zeros((n, 7), np.float32)
        self._l.emul_scatter(self._h, rays.ctypes.data, index.ctypes.data, t.ctypes.data, u4.ctypes.data, n, out.ctypes.data)
        return out

    def render_tiles(self, cam22, params, threads=8, seg_counts=None, accel=0):
        nlt = rt.local_tile_count(params.width, params.height, params.rank, max(1, params.world_size), params.tile_w or 8, params.tile_h or 8)
        out = np.zeros((nlt, 64, 3), np.float32)
        st = np.zeros(4, np.uint64)
        cam = np.ascontiguousarray(cam22, np.float32)
        rc = self._l.emul_render_tiles(self._h, cam.ctypes.data, C.byref(params), out.ctypes.data, st.ctypes.data, threads,
                                          seg_counts.ctypes.data if seg_counts is not None else None, accel)
        assert rc == 0
        return out, {"rays": int(st[0]), "samples": int(st[1]), "exact_tests": int(st[2]), "passes": int(st[3])}

    def render(self, cam22, W, H, spp, depth, seed=0, threads=8, run_len=0, tile_w=8, tile_h=8, sample_offset=0, accel=0):
        from np_mirror import deinterleave
        p = rt.make_params(W, H, spp, depth, seed, 0.001, 0, 1, tile_w, tile_h, run_len, sample_offset)
        tiles, st = self.render_tiles(cam22, p, threads, accel=accel)
        return deinterleave(tiles[None], W, H, 1, tile_w, tile_h), st


def primary_rays(cam22, W, H, xys, seed=0):
    xys = np.ascontiguousarray(xys, np.int32).reshape(-1, 3)
    out = np.zeros((xys.shape[0], 6), np.float32)
    cam = np.ascontiguousarray(cam22, np.float32)
    lib().emul_primary(cam.ctypes.data, W, H, seed, xys.ctypes.data, xys.shape[0], out.ctypes.data)
    return out


def philox(seed, ctr4):
    ctr4 = np.ascontiguousarray(ctr4, np.uint32).reshape(-1, 4)
    out = np.zeros_like(ctr4)
    lib().emul_philox(seed, ctr4.ctypes.data, ctr4.shape[0], out.ctypes.data)
    return out
