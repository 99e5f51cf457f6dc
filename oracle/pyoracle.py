"""ctypes binding of oracle/liboracle.so (oracle/cpu_ref.cpp) and, when built, oracle/_ref/libref_harness.so.

TEST INFRASTRUCTURE ONLY — import from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg, never from the product package.
"""
from __future__ import annotations

import ctypes as C
import hashlib
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liboracle.so")
REF_HARNESS_PATH = os.path.join(HERE, "_ref", "libref_harness.so")
REF_BINARY_PATH = os.path.join(HERE, "_ref", "rtiow_ref")

_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")

SCENE_FIELDS = ("cx", "cy", "cz", "radius", "mat_r", "mat_g", "mat_b", "roughness", "metallic", "ior", "mat_type")


def build(ref: bool = True) -> None:
    """Compile liboracle.so and, when /root/reference is present, oracle/_ref/*."""
    subprocess.run(["make", "-C", HERE, "all"], check=True, capture_output=True)
    if ref and os.path.isdir("/root/reference/RayTracingInOneWeekend"):
        subprocess.run(["make", "-C", HERE, "ref"], check=True, capture_output=True)


def _load() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        build(ref=False)
    lib = C.CDLL(LIB_PATH)
    lib.oracle_create.restype = C.c_void_p
    lib.oracle_destroy.argtypes = [C.c_void_p]
    lib.oracle_discard.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64]
    lib.oracle_random_scene.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float]
    lib.oracle_random_scene.restype = C.c_int
    lib.oracle_scene_size.argtypes = [C.c_void_p]
    lib.oracle_scene_size.restype = C.c_int
    lib.oracle_set_scene.argtypes = [C.c_void_p, C.c_int] + [_f32p] * 10 + [_i32p]
    lib.oracle_get_scene.argtypes = [C.c_void_p] + [_f32p] * 10 + [_i32p]
    lib.oracle_camera.argtypes = [_f32p, _f32p, _f32p, C.c_float, C.c_float, C.c_float, C.c_float, _f32p]
    lib.oracle_intersect_batch.argtypes = [C.c_void_p, _f32p, C.c_uint64, C.c_float, C.c_float, _u8p, _i32p, _f32p]
    lib.oracle_disc_pass.argtypes = [C.c_void_p, _f32p, C.c_uint64, _u8p]
    lib.oracle_sphere_hit_batch.argtypes = [C.c_void_p, _f32p, _i32p, C.c_uint64, C.c_float, C.c_float, _f32p]
    lib.oracle_scatter_batch.argtypes = [C.c_void_p, _f32p, _i32p, _f32p, _f32p, C.c_uint64, _f32p]
    lib.oracle_sample.argtypes = [C.c_void_p, C.c_int, C.c_uint64, _f32p]
    lib.oracle_render_rows.argtypes = [C.c_void_p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_void_p, C.c_void_p]
    lib.oracle_render_rows.restype = C.c_uint64
    lib.oracle_depth_outs.argtypes = [C.c_void_p]
    lib.oracle_depth_outs.restype = C.c_uint64
    lib.oracle_quantise.argtypes = [_f32p, C.c_uint64, C.c_int, _u8p]
    lib.oracle_primary_batch.argtypes = [_f32p, C.c_int, C.c_int, _i32p, _f32p, _f32p, C.c_uint64, _f32p]
    return lib


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


def stock_camera(aspect: float) -> np.ndarray:
    """The fixed camera of main.cpp:48-54 for a given aspect ratio -> 22 floats."""
    return camera((13.0, 2.0, 3.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 20.0, aspect, 0.1, 10.0)


def camera(look_from, look_at, vup, vfov, aspect, aperture, focus) -> np.ndarray:
    out = np.zeros(22, np.float32)
    lib().oracle_camera(np.asarray(look_from, np.float32), np.asarray(look_at, np.float32), np.asarray(vup, np.float32),
                        vfov, aspect, aperture, focus, out)
    return out


def cli_dims(argv):
    """main.cpp:23-41 -> (W, H, spp, depth, aspect) with fp32 arithmetic where the reference has it."""
    W = int(argv[0]) if len(argv) > 0 else 400
    aspect = np.float32(3.0) / np.float32(2.0)
    if len(argv) > 1:
        H = int(argv[1])
        aspect = np.float32(W) / np.float32(H)
    else:
        H = int(np.float32(W) / aspect)
    spp = int(argv[2]) if len(argv) > 2 else 100
    depth = int(argv[3]) if len(argv) > 3 else 50
    return W, H, spp, depth, float(aspect)


class Oracle:
    """One reference 'process': two default-seeded mt19937 streams + a scene."""

    def __init__(self):
        self._l = lib()
        self._h = C.c_void_p(self._l.oracle_create())

    def __del__(self):
        try:
            self._l.oracle_destroy(self._h)
        except Exception:
            pass

    def discard(self, n0: int, n1: int) -> None:
        self._l.oracle_discard(self._h, n0, n1)

    def random_scene(self, k: int = 11, diffuse_below: float = 0.8, metal_below: float = 0.95) -> int:
        return self._l.oracle_random_scene(self._h, k, diffuse_below, metal_below)

    @property
    def n(self) -> int:
        return self._l.oracle_scene_size(self._h)

    def get_scene(self) -> dict:
        n = self.n
        arrs = [np.zeros(n, np.float32) for _ in range(10)] + [np.zeros(n, np.int32)]
        self._l.oracle_get_scene(self._h, *arrs)
        return dict(zip(SCENE_FIELDS, arrs))

    def set_scene(self, scene: dict) -> None:
        arrs = [np.ascontiguousarray(scene[k], np.float32) for k in SCENE_FIELDS[:10]]
        arrs.append(np.ascontiguousarray(scene["mat_type"], np.int32))
        self._l.oracle_set_scene(self._h, len(arrs[0]), *arrs)

    def intersect(self, rays: np.ndarray, t_min: float = 0.001, t_max: float = float("inf")):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        n = rays.shape[0]
        hit = np.zeros(n, np.uint8)
        idx = np.zeros(n, np.int32)
        t = np.zeros(n, np.float32)
        self._l.oracle_intersect_batch(self._h, rays, n, t_min, t_max, hit, idx, t)
        return hit, idx, t

    def disc_pass(self, rays) -> np.ndarray:
        """[n_rays, n_spheres] uint8: 1 where the reference's discriminant test lets the sphere through."""
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        out = np.zeros((rays.shape[0], self.n), np.uint8)
        self._l.oracle_disc_pass(self._h, rays, rays.shape[0], out.reshape(-1))
        return out

    def sphere_hit(self, rays, sphere, t_min=0.001, t_max=float("inf")):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        sphere = np.ascontiguousarray(sphere, np.int32)
        out = np.zeros((rays.shape[0], 9), np.float32)
        self._l.oracle_sphere_hit_batch(self._h, rays, sphere, rays.shape[0], t_min, t_max, out)
        return out

    def scatter(self, rays, sphere, t, rnd7):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        out = np.zeros((rays.shape[0], 7), np.float32)
        self._l.oracle_scatter_batch(self._h, rays, np.ascontiguousarray(sphere, np.int32),
                                     np.ascontiguousarray(t, np.float32), np.ascontiguousarray(rnd7, np.float32),
                                     rays.shape[0], out)
        return out

    def sample(self, kind: int, n: int) -> np.ndarray:
        out = np.zeros((n, 3), np.float32)
        self._l.oracle_sample(self._h, kind, n, out)
        return out

    def render_rows(self, cam22, W, H, spp, depth, row_begin=0, row_end=None, want_u8=True):
        row_end = H if row_end is None else row_end
        rows = row_end - row_begin
        sums = np.zeros((rows, W, 3), np.float32)
        u8 = np.zeros((rows, W, 3), np.uint8) if want_u8 else None
        seg = self._l.oracle_render_rows(self._h, np.ascontiguousarray(cam22, np.float32), W, H, spp, depth, row_begin,
                                         row_end, sums.ctypes.data, u8.ctypes.data if want_u8 else None)
        return sums, u8, int(seg)

    def depth_outs(self) -> int:
        return int(self._l.oracle_depth_outs(self._h))


def primary_rays(cam22, W, H, xy, xi, disk) -> np.ndarray:
    """main.cpp:69-71 + Camera::get_ray for pixel (x, row-from-top) pairs with given jitter and lens samples -> [n, 6]."""
    xy = np.ascontiguousarray(xy, np.int32).reshape(-1, 2)
    out = np.zeros((xy.shape[0], 6), np.float32)
    lib().oracle_primary_batch(np.ascontiguousarray(cam22, np.float32), W, H, xy, np.ascontiguousarray(xi, np.float32).reshape(-1, 2),
                               np.ascontiguousarray(disk, np.float32).reshape(-1, 2), xy.shape[0], out)
    return out


def quantise(sums: np.ndarray, spp: int) -> np.ndarray:
    """write_color_binary bytes (Color.hpp:28-48) for an array of colour sums."""
    flat = np.ascontiguousarray(sums, np.float32).reshape(-1)
    out = np.zeros(flat.shape[0], np.uint8)
    lib().oracle_quantise(flat, flat.shape[0], spp, out)
    return out.reshape(sums.shape)


def ppm_bytes(u8: np.ndarray) -> bytes:
    """main.cpp:59-60 header + pixel bytes."""
    H, W, _ = u8.shape
    return b"P6\n%d %d\n255\n" % (W, H) + u8.tobytes()


def reference_main_ppm_sha256(argv) -> str:
    """What the reference executable writes for this argv, restated: scene, camera, render, PPM."""
    W, H, spp, depth, aspect = cli_dims(argv)
    o = Oracle()
    o.random_scene()
    cam = stock_camera(aspect)
    _, u8, _ = o.render_rows(cam, W, H, spp, depth)
    return hashlib.sha256(ppm_bytes(u8)).hexdigest()


def psnr_u8(a: np.ndarray, b: np.ndarray) -> float:
    d = a.astype(np.float64) - b.astype(np.float64)
    mse = float(np.mean(d * d))
    return float("inf") if mse == 0 else 10.0 * np.log10(255.0 * 255.0 / mse)
