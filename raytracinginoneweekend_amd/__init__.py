"""raytracinginoneweekend_amd — MI355X-native path tracer behind the RayTracingInOneWeekend scene API.

The product is native: `librtk.so` (hand-written HIP kernels + the C ABI of include/rtk.h) and the C++
host scene API under include/rtiow/.  This Python package is a thin ctypes view of both, for tests and
bench.py; it holds no rendering logic of its own and has NO CPU fallback: if librtk.so is missing or no
gfx950 device is visible, every compute call raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RTK_LIB", os.path.join(HERE, "librtk.so"))   # RTK_LIB: A/B builds of the kernel (tools/)
HOST_LIB_PATH = os.path.join(HERE, "librtiow_host.so")

SCENE_FIELDS = ("cx", "cy", "cz", "radius", "mat_r", "mat_g", "mat_b", "roughness", "metallic", "ior", "mat_type")

# Every symbol include/rtk.h declares (tests check that the library exports all of them).
C_ABI_SYMBOLS = (
    "rtk_device_count", "rtk_last_error", "rtk_context_create", "rtk_context_destroy", "rtk_context_set_stream",
    "rtk_set_scene", "rtk_local_tile_count", "rtk_render_tiles_device", "rtk_deinterleave_device", "rtk_render",
    "rtk_quantise_device", "rtk_intersect_batch", "rtk_intersect_batch_device", "rtk_scatter_batch",
    "rtk_primary_rays", "rtk_philox_batch", "rtk_synchronize", "rtk_reset_stats", "rtk_get_stats",
    "rtk_run_schedule", "rtk_deinterleave_u8_device", "rtk_render_u8", "rtk_context_set_accel", "rtk_accel_in_use",
    "rtk_render_tiles_device_noalloc", "rtk_scene_layout", "rtk_sample_batch",
    "rtk_multi_create", "rtk_multi_destroy", "rtk_multi_device_count", "rtk_multi_uses_rccl", "rtk_multi_set_accel", "rtk_multi_set_scene",
    "rtk_multi_render",
)
ACCEL_LIST, ACCEL_GRID = 0, 1


class RtkError(RuntimeError):
    pass


class rtk_scene(C.Structure):
    _fields_ = [("n", C.c_int32)] + [(k, C.POINTER(C.c_float)) for k in SCENE_FIELDS[:10]] + [("mat_type", C.POINTER(C.c_int32))]


class rtk_camera(C.Structure):
    _fields_ = [("f", C.c_float * 22)]


class rtk_params(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("spp", C.c_int32), ("max_depth", C.c_int32),
                ("seed", C.c_uint64), ("t_min", C.c_float), ("rank", C.c_int32), ("world_size", C.c_int32),
                ("tile_w", C.c_int32), ("tile_h", C.c_int32), ("run_len", C.c_int32), ("sample_offset", C.c_int32)]


class rtk_stats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("samples", C.c_uint64), ("sphere_tests", C.c_uint64), ("exact_tests", C.c_uint64),
                ("lane_slots", C.c_uint64), ("kernel_ms", C.c_float), ("total_ms", C.c_float), ("frames", C.c_uint32),
                ("frames_timed", C.c_uint32), ("max_wave_passes", C.c_uint64), ("waves", C.c_uint64),
                ("drain_frac", C.c_float), ("reserved_", C.c_float)]

    def as_dict(self) -> dict:
        return {k: getattr(self, k) for k, _ in self._fields_}


_lib = None
_host = None


def lib() -> C.CDLL:
    """librtk.so; raises if it has not been built (python -m raytracinginoneweekend_amd.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RtkError(f"{LIB_PATH} is missing: build it with `python -m raytracinginoneweekend_amd.build` "
                           "(there is no CPU fallback)")
        # One HIP runtime per process: torch bundles its own libamdhip64.so.7 / libhsa-runtime64.so.1.  If torch
        # is going to be used in this process (bench.py, tests) it must be loaded BEFORE librtk.so so that the
        # dynamic loader resolves librtk's libamdhip64.so.7 to the copy torch already mapped; the other order
        # leaves torch without devices ("No HIP GPUs are available").
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        l = C.CDLL(LIB_PATH)
        vp, i32, i64, u64, f32, sz = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_float, C.c_size_t
        l.rtk_device_count.restype = C.c_int
        l.rtk_last_error.restype = C.c_char_p
        l.rtk_context_create.argtypes = [C.c_int, C.POINTER(vp)]
        l.rtk_context_destroy.argtypes = [vp]
        l.rtk_context_destroy.restype = None
        l.rtk_context_set_stream.argtypes = [vp, vp]
        l.rtk_set_scene.argtypes = [vp, C.POINTER(rtk_scene)]
        l.rtk_context_set_accel.argtypes = [vp, i32]
        l.rtk_accel_in_use.argtypes = [vp]
        l.rtk_local_tile_count.argtypes = [i32, i32, i32, i32, i32, i32]
        l.rtk_local_tile_count.restype = i64
        l.rtk_render_tiles_device.argtypes = [vp, C.POINTER(rtk_camera), C.POINTER(rtk_params), vp, C.POINTER(rtk_stats)]
        l.rtk_scene_layout.argtypes = [vp, vp]
        l.rtk_render_tiles_device_noalloc.argtypes = [vp, C.POINTER(rtk_camera), C.POINTER(rtk_params), vp, C.POINTER(rtk_stats)]
        l.rtk_deinterleave_device.argtypes = [vp, vp, i64, C.POINTER(rtk_params), vp]
        l.rtk_deinterleave_u8_device.argtypes = [vp, vp, i64, C.POINTER(rtk_params), vp]
        l.rtk_render.argtypes = [vp, C.POINTER(rtk_camera), C.POINTER(rtk_params), vp, C.POINTER(rtk_stats)]
        l.rtk_render_u8.argtypes = [vp, C.POINTER(rtk_camera), C.POINTER(rtk_params), vp, C.POINTER(rtk_stats)]
        l.rtk_quantise_device.argtypes = [vp, vp, i64, i32, vp]
        l.rtk_intersect_batch.argtypes = [vp, vp, sz, f32, f32, vp, vp, vp]
        l.rtk_intersect_batch_device.argtypes = [vp, vp, sz, f32, f32, vp, vp, vp]
        l.rtk_scatter_batch.argtypes = [vp, vp, vp, vp, vp, sz, vp]
        l.rtk_primary_rays.argtypes = [vp, C.POINTER(rtk_camera), C.POINTER(rtk_params), vp, sz, vp]
        l.rtk_philox_batch.argtypes = [vp, u64, vp, sz, vp]
        l.rtk_sample_batch.argtypes = [vp, i32, vp, sz, vp]
        l.rtk_multi_create.argtypes = [vp, i32, C.POINTER(vp)]
        l.rtk_multi_destroy.argtypes = [vp]
        l.rtk_multi_destroy.restype = None
        l.rtk_multi_device_count.argtypes = [vp]
        l.rtk_multi_uses_rccl.argtypes = [vp]
        l.rtk_multi_set_accel.argtypes = [vp, i32]
        l.rtk_multi_set_scene.argtypes = [vp, C.POINTER(rtk_scene)]
        l.rtk_multi_render.argtypes = [vp, C.POINTER(rtk_camera), C.POINTER(rtk_params), vp, vp, vp]
        l.rtk_synchronize.argtypes = [vp]
        l.rtk_reset_stats.argtypes = [vp]
        l.rtk_get_stats.argtypes = [vp, C.POINTER(rtk_stats)]
        l.rtk_run_schedule.argtypes = [i32, i32, i32, i32, vp, i32]
        _lib = l
    return _lib


def host_lib() -> C.CDLL:
    """librtiow_host.so: C view of the C++ host scene API (CPU only)."""
    global _host
    if _host is None:
        if not os.path.exists(HOST_LIB_PATH):
            raise RtkError(f"{HOST_LIB_PATH} is missing: build it with `python -m raytracinginoneweekend_amd.build`")
        h = C.CDLL(HOST_LIB_PATH)
        fp = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
        ip = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
        up = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
        h.rtiow_host_random_scene.argtypes = [C.c_int] + [fp] * 10 + [ip]
        h.rtiow_host_scaled_scene.argtypes = [C.c_int, C.c_float, C.c_float, C.c_uint32, C.c_int] + [fp] * 10 + [ip]
        h.rtiow_host_camera.argtypes = [fp, fp, fp, C.c_float, C.c_float, C.c_float, C.c_float, fp]
        h.rtiow_host_flatten_nested.argtypes = [C.c_int, fp, fp, fp, fp, C.c_int, C.c_int] + [fp] * 10 + [ip]
        h.rtiow_host_list_hit.argtypes = [C.c_int, fp, fp, fp, fp, C.c_int, fp, C.c_uint64, C.c_float, C.c_float, up, ip, fp]
        h.rtiow_host_write_color_binary.argtypes = [fp, C.c_uint64, C.c_int, up]
        h.rtiow_host_write_ppm.argtypes = [C.c_int, C.c_int, fp, C.c_int, up, C.c_int64]
        h.rtiow_host_write_ppm.restype = C.c_int64
        h.rtiow_host_cpu_render.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, up]
        _host = h
    return _host


def _check(status: int, what: str) -> None:
    if status != 0:
        raise RtkError(f"{what} failed ({status}): {lib().rtk_last_error().decode()}")


# ---- host scene API helpers (C++ classes behind librtiow_host.so) ---------------------------------

def _scene_arrays(cap: int):
    return [np.zeros(cap, np.float32) for _ in range(10)] + [np.zeros(cap, np.int32)]


def scaled_scene(k: int = 11, diffuse_below: float = 0.8, metal_below: float = 0.95, seed: int = 5489) -> dict:
    """random_scene() generalised (grid half-width k, material thresholds); defaults give the stock scene."""
    cap = 4 * k * k + 8
    arrs = _scene_arrays(cap)
    n = host_lib().rtiow_host_scaled_scene(k, diffuse_below, metal_below, seed, cap, *arrs)
    if n < 0:
        raise RtkError("scene capacity too small")
    return {f: a[:n].copy() for f, a in zip(SCENE_FIELDS, arrs)}


def flatten_nested(cx, cy, cz, radius, nest: int) -> dict:
    """A user scene built from nested HittableLists with the host classes (every `nest` spheres in an inner list, some
    wrapped two levels deeper) and flattened by rtiow::flatten: the arrays rtk_set_scene takes, in the reference's visiting order."""
    n = len(cx)
    arrs = _scene_arrays(n + 8)
    f = lambda a: np.ascontiguousarray(a, np.float32)   # noqa: E731
    got = host_lib().rtiow_host_flatten_nested(n, f(cx), f(cy), f(cz), f(radius), nest, n + 8, *arrs)
    if got < 0:
        raise RtkError("scene capacity too small")
    return {k: a[:got].copy() for k, a in zip(SCENE_FIELDS, arrs)}


def stock_scene() -> dict:
    """The book-1 final scene exactly as the reference builds it under libstdc++ (484 spheres)."""
    return scaled_scene()


def mixed_radii_scene(n: int = 200, seed: int = 1, extent: float = 8.0) -> dict:
    """A user scene (tests, tools): the ground sphere plus n - 1 spheres of varied radius resting on it, materials mixed.
    No two small spheres share (cy, radius), so the matrix-pipe layout builds GENERAL words (and a few class words when n
    is large): the loop the stock scene never enters."""
    rng = np.random.default_rng(seed)
    m = n - 1
    r = rng.uniform(0.15, 0.6, m).astype(np.float32)
    s = {k: np.zeros(n, np.float32) for k in SCENE_FIELDS[:10]}
    s["mat_type"] = np.ones(n, np.int32)
    s["cx"][1:] = rng.uniform(-extent, extent, m); s["cz"][1:] = rng.uniform(-extent, extent, m)
    s["cy"][1:] = r; s["radius"][1:] = r
    s["cy"][0] = -1000.0; s["radius"][0] = 1000.0
    for k in ("mat_r", "mat_g", "mat_b"):
        s[k][:] = rng.uniform(0.2, 0.95, n)
    kind = rng.uniform(size=n)
    s["mat_type"][kind > 0.6] = 2; s["mat_type"][kind > 0.85] = 3
    s["mat_type"][0] = 1
    s["roughness"][:] = np.where(s["mat_type"] == 2, rng.uniform(0, 0.5, n), 1.0)
    s["metallic"][:] = np.where(s["mat_type"] == 2, 1.0, 0.0)
    s["ior"][:] = np.where(s["mat_type"] == 3, 1.5, 0.0)
    glass = s["mat_type"] == 3
    for k in ("mat_r", "mat_g", "mat_b"):
        s[k][glass] = 1.0
    return s


def camera(look_from, look_at, vup, vfov, aspect, aperture, focus) -> np.ndarray:
    out = np.zeros(22, np.float32)
    host_lib().rtiow_host_camera(np.asarray(look_from, np.float32), np.asarray(look_at, np.float32),
                                 np.asarray(vup, np.float32), vfov, aspect, aperture, focus, out)
    return out


def stock_camera(aspect: float) -> np.ndarray:
    """Camera of main.cpp:48-54."""
    return camera((13.0, 2.0, 3.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 20.0, aspect, 0.1, 10.0)


def run_schedule(spp: int, W: int, H: int, run_len: int = 0) -> np.ndarray:
    """[n_runs, 2] (first sample, length): how a render of this size splits a pixel's samples (host only)."""
    n = lib().rtk_run_schedule(spp, W, H, run_len, None, 0)
    if n < 0:
        raise RtkError(lib().rtk_last_error().decode())
    out = np.zeros((n, 2), np.uint32)
    _check(min(0, lib().rtk_run_schedule(spp, W, H, run_len, out.ctypes.data, n)), "rtk_run_schedule")
    return out


def local_tile_count(W, H, rank=0, world=1, tile_w=8, tile_h=8) -> int:
    return int(lib().rtk_local_tile_count(W, H, tile_w, tile_h, rank, world))


def _cam_struct(cam22) -> rtk_camera:
    c = rtk_camera()
    a = np.ascontiguousarray(cam22, np.float32)
    assert a.shape == (22,)
    C.memmove(C.byref(c), a.ctypes.data, 88)
    return c


def make_params(W, H, spp, depth, seed=0, t_min=0.001, rank=0, world=1, tile_w=8, tile_h=8, run_len=0, sample_offset=0) -> rtk_params:
    return rtk_params(W, H, spp, depth, seed, t_min, rank, world, tile_w, tile_h, run_len, sample_offset)


def _scene_struct(scene: dict):
    """rtk_scene view of a scene dict -> (struct, arrays it points into: keep them alive for the call)."""
    arrs = {k: np.ascontiguousarray(scene[k], np.float32) for k in SCENE_FIELDS[:10]}
    arrs["mat_type"] = np.ascontiguousarray(scene["mat_type"], np.int32)
    s = rtk_scene()
    s.n = len(arrs["cx"])
    for k in SCENE_FIELDS[:10]:
        setattr(s, k, arrs[k].ctypes.data_as(C.POINTER(C.c_float)))
    s.mat_type = arrs["mat_type"].ctypes.data_as(C.POINTER(C.c_int32))
    return s, arrs


class Multi:
    """rtk_multi: one process driving several GPUs (ranks = devices, in order; a device may be listed more than once)."""

    def __init__(self, devices):
        self._l = lib()
        self.devices = [int(d) for d in devices]
        arr = (C.c_int32 * len(self.devices))(*self.devices)
        h = C.c_void_p()
        _check(self._l.rtk_multi_create(arr, len(self.devices), C.byref(h)), "rtk_multi_create")
        self._h = h
        self._scene_keep = None

    def close(self):
        if self._h:
            self._l.rtk_multi_destroy(self._h)
            self._h = None

    @property
    def uses_rccl(self) -> bool:
        return bool(self._l.rtk_multi_uses_rccl(self._h))

    def set_accel(self, mode: int):
        _check(self._l.rtk_multi_set_accel(self._h, mode), "rtk_multi_set_accel")

    def set_scene(self, scene: dict):
        s, keep = _scene_struct(scene)
        self._scene_keep = keep
        _check(self._l.rtk_multi_set_scene(self._h, C.byref(s)), "rtk_multi_set_scene")

    def render(self, cam22, W, H, spp, depth, seed=0, want_u8=False, t_min=0.001, tile_w=8, tile_h=8, run_len=0, sample_offset=0):
        """-> (float32 [H, W, 3] colour sums, uint8 [H, W, 3] or None, list of per-device stats dicts)."""
        out = np.zeros((H, W, 3), np.float32)
        u8 = np.zeros((H, W, 3), np.uint8) if want_u8 else None
        st = (rtk_stats * len(self.devices))()
        p = make_params(W, H, spp, depth, seed, t_min, 0, 1, tile_w, tile_h, run_len, sample_offset)
        cam = _cam_struct(cam22)
        _check(self._l.rtk_multi_render(self._h, C.byref(cam), C.byref(p), out.ctypes.data, u8.ctypes.data if want_u8 else None, st), "rtk_multi_render")
        return out, u8, [x.as_dict() for x in st]


class Context:
    """rtk_context on one device."""

    def __init__(self, device: int = 0):
        self._l = lib()
        h = C.c_void_p()
        _check(self._l.rtk_context_create(device, C.byref(h)), "rtk_context_create")
        self._h = h
        self._scene_keep = None

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._l.rtk_context_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_handle: int) -> None:
        _check(self._l.rtk_context_set_stream(self._h, C.c_void_p(stream_handle)), "rtk_context_set_stream")

    def set_scene(self, scene: dict) -> None:
        s, _keep = _scene_struct(scene)
        _check(self._l.rtk_set_scene(self._h, C.byref(s)), "rtk_set_scene")
        self.n = s.n

    def set_accel(self, mode: int) -> int:
        """ACCEL_LIST (default) or ACCEL_GRID; returns the mode the next call will really use for the current scene."""
        _check(self._l.rtk_context_set_accel(self._h, mode), "rtk_context_set_accel")
        return self._l.rtk_accel_in_use(self._h)

    def accel_in_use(self) -> int:
        return self._l.rtk_accel_in_use(self._h)

    def reset_stats(self) -> None:
        _check(self._l.rtk_reset_stats(self._h), "rtk_reset_stats")

    def get_stats(self) -> dict:
        st = rtk_stats()
        _check(self._l.rtk_get_stats(self._h, C.byref(st)), "rtk_get_stats")
        return st.as_dict()

    def synchronize(self) -> None:
        _check(self._l.rtk_synchronize(self._h), "rtk_synchronize")

    def render(self, cam22, W, H, spp, depth, seed=0, t_min=0.001, tile_w=8, tile_h=8, run_len=0, sample_offset=0):
        """Whole image -> (float32 [H, W, 3] colour sums, stats dict)."""
        out = np.zeros((H, W, 3), np.float32)
        st = rtk_stats()
        p = make_params(W, H, spp, depth, seed, t_min, 0, 1, tile_w, tile_h, run_len, sample_offset)
        cam = _cam_struct(cam22)
        _check(self._l.rtk_render(self._h, C.byref(cam), C.byref(p), out.ctypes.data, C.byref(st)), "rtk_render")
        return out, st.as_dict()

    def render_u8(self, cam22, W, H, spp, depth, seed=0, t_min=0.001, tile_w=8, tile_h=8, run_len=0):
        """Whole image, quantised on the device -> (uint8 [H, W, 3] = the PPM payload, stats dict)."""
        out = np.zeros((H, W, 3), np.uint8)
        st = rtk_stats()
        p = make_params(W, H, spp, depth, seed, t_min, 0, 1, tile_w, tile_h, run_len, 0)
        cam = _cam_struct(cam22)
        _check(self._l.rtk_render_u8(self._h, C.byref(cam), C.byref(p), out.ctypes.data, C.byref(st)), "rtk_render_u8")
        return out, st.as_dict()

    def render_tiles_device(self, cam22, params: rtk_params, d_tile_sums_ptr: int, want_stats: bool = False):
        st = rtk_stats()
        cam = _cam_struct(cam22)
        _check(self._l.rtk_render_tiles_device(self._h, C.byref(cam), C.byref(params), C.c_void_p(d_tile_sums_ptr),
                                               C.byref(st) if want_stats else None), "rtk_render_tiles_device")
        return st.as_dict() if want_stats else None

    def scene_layout(self) -> dict:
        a = (C.c_int32 * 4)()
        _check(self._l.rtk_scene_layout(self._h, a), "rtk_scene_layout")
        return {"spheres": a[0], "class_words": a[1], "lds_resident": bool(a[2]), "lds_bytes_per_workgroup": a[3]}

    def render_tiles_device_noalloc(self, cam22, params: rtk_params, d_tile_sums_ptr: int, want_stats: bool = False):
        """rtk_render_tiles_device with the kernels' working memory in the library's static device array (rtk.h)."""
        st = rtk_stats()
        cam = _cam_struct(cam22)
        _check(self._l.rtk_render_tiles_device_noalloc(self._h, C.byref(cam), C.byref(params), C.c_void_p(d_tile_sums_ptr),
                                                       C.byref(st) if want_stats else None), "rtk_render_tiles_device_noalloc")
        return st.as_dict() if want_stats else None

    def deinterleave_device(self, d_gathered_ptr: int, tiles_per_rank_padded: int, params: rtk_params, d_image_ptr: int):
        _check(self._l.rtk_deinterleave_device(self._h, C.c_void_p(d_gathered_ptr), tiles_per_rank_padded, C.byref(params),
                                               C.c_void_p(d_image_ptr)), "rtk_deinterleave_device")

    def deinterleave_u8_device(self, d_gathered_ptr: int, tiles_per_rank_padded: int, params: rtk_params, d_image_ptr: int):
        _check(self._l.rtk_deinterleave_u8_device(self._h, C.c_void_p(d_gathered_ptr), tiles_per_rank_padded, C.byref(params),
                                                  C.c_void_p(d_image_ptr)), "rtk_deinterleave_u8_device")

    def quantise_device(self, d_sum_ptr: int, n_pixels: int, spp: int, d_u8_ptr: int):
        _check(self._l.rtk_quantise_device(self._h, C.c_void_p(d_sum_ptr), n_pixels, spp, C.c_void_p(d_u8_ptr)),
               "rtk_quantise_device")

    def intersect(self, rays, t_min=0.001, t_max=float("inf")):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        n = rays.shape[0]
        hit = np.zeros(n, np.uint8)
        idx = np.zeros(n, np.int32)
        t = np.zeros(n, np.float32)
        _check(self._l.rtk_intersect_batch(self._h, rays.ctypes.data, n, t_min, t_max, hit.ctypes.data, idx.ctypes.data,
                                           t.ctypes.data), "rtk_intersect_batch")
        return hit, idx, t

    def intersect_device(self, d_rays_ptr, n, d_hit_ptr, d_idx_ptr, d_t_ptr, t_min=0.001, t_max=float("inf")):
        _check(self._l.rtk_intersect_batch_device(self._h, C.c_void_p(d_rays_ptr), n, t_min, t_max, C.c_void_p(d_hit_ptr),
                                                  C.c_void_p(d_idx_ptr), C.c_void_p(d_t_ptr)), "rtk_intersect_batch_device")

    def scatter(self, rays, index, t, u4):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        n = rays.shape[0]
        index = np.ascontiguousarray(index, np.int32)
        t = np.ascontiguousarray(t, np.float32)
        u4 = np.ascontiguousarray(u4, np.float32).reshape(n, 4)
        out = np.zeros((n, 7), np.float32)
        _check(self._l.rtk_scatter_batch(self._h, rays.ctypes.data, index.ctypes.data, t.ctypes.data, u4.ctypes.data, n,
                                         out.ctypes.data), "rtk_scatter_batch")
        return out

    def primary_rays(self, cam22, W, H, xys, seed=0):
        xys = np.ascontiguousarray(xys, np.int32).reshape(-1, 3)
        out = np.zeros((xys.shape[0], 6), np.float32)
        p = make_params(W, H, 1, 1, seed)
        cam = _cam_struct(cam22)
        _check(self._l.rtk_primary_rays(self._h, C.byref(cam), C.byref(p), xys.ctypes.data, xys.shape[0], out.ctypes.data),
               "rtk_primary_rays")
        return out

    def sample(self, kind: int, u4):
        """The kernels' direct sampling maps: kind 0 ball, 1 unit sphere, 2 disk; u4 [n, 4] uniforms -> [n, 3]."""
        u4 = np.ascontiguousarray(u4, np.float32).reshape(-1, 4)
        out = np.zeros((u4.shape[0], 3), np.float32)
        _check(self._l.rtk_sample_batch(self._h, kind, u4.ctypes.data, u4.shape[0], out.ctypes.data), "rtk_sample_batch")
        return out

    def philox(self, seed: int, ctr4):
        ctr4 = np.ascontiguousarray(ctr4, np.uint32).reshape(-1, 4)
        out = np.zeros_like(ctr4)
        _check(self._l.rtk_philox_batch(self._h, seed, ctr4.ctypes.data, ctr4.shape[0], out.ctypes.data), "rtk_philox_batch")
        return out
