"""ctypes binding of libmi355rt.so (include/mi355rt.h) -- plumbing for tests/ and bench.py.
No CPU fallback: raises if the library is missing or no HIP device is present."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "libmi355rt.so")

RT_OK, RT_ERR_ARG, RT_ERR_NO_DEVICE = 0, -2001, -2003
RT_MODE_BRUTE, RT_MODE_BINNED = 0, 1

SPHERE_DTYPE = np.dtype([("r", "<f4"), ("b", "<f4"), ("g", "<f4"), ("radius", "<f4"),
                         ("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("idx", "<i4")])   # sphere.cuh:28-32


class RtStats(C.Structure):
    _fields_ = [("ms_render", C.c_float), ("mode", C.c_uint32), ("sphere_tests", C.c_uint64)]


EXPORTS = ["rt_create", "rt_destroy", "rt_set_spheres", "rt_set_mode", "rt_render", "rt_render_rows", "rt_render_repeat",
           "rt_init_shifts", "rt_anim_init", "rt_anim_axis_move", "rt_anim_curve_move", "rt_anim_update_speed_angle",
           "rt_anim_get_state", "rt_anim_loop", "rt_get_stats", "rt_version"]

_lib = None


def load_library(path: str = LIB_PATH) -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    try:                                                 # one HIP runtime per process: torch's bundled copy first (see mi355cd.load_library)
        import torch  # noqa: F401
    except Exception:
        pass
    path = os.environ.get("MI355RT_LIB", path)          # A/B runs of two builds (tools/)
    if not os.path.exists(path):
        raise RuntimeError(f"{path} not built: run __graft_entry__.build() (there is no CPU fallback)")
    lib = C.CDLL(path)
    vp = C.c_void_p
    lib.rt_create.argtypes = [C.POINTER(vp), vp, C.c_int32, C.c_int32]
    lib.rt_destroy.argtypes = [vp]; lib.rt_destroy.restype = None
    lib.rt_set_spheres.argtypes = [vp, vp]
    lib.rt_set_mode.argtypes = [vp, C.c_int]
    lib.rt_render.argtypes = [vp, vp, C.c_int32, C.c_int32, vp]
    lib.rt_render_rows.argtypes = [vp, vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, vp]
    lib.rt_render_repeat.argtypes = [vp, vp, C.c_int32, C.c_int32, C.c_int32, vp]
    lib.rt_init_shifts.argtypes = [C.c_int32, vp, vp]
    lib.rt_anim_init.argtypes = [vp]
    lib.rt_anim_axis_move.argtypes = [vp, C.c_int32]
    lib.rt_anim_curve_move.argtypes = [vp]
    lib.rt_anim_update_speed_angle.argtypes = [vp, C.c_int32, C.c_int32]
    lib.rt_anim_get_state.argtypes = [vp, vp, vp, vp]
    lib.rt_anim_loop.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, vp]
    lib.rt_get_stats.argtypes = [vp, C.POINTER(RtStats)]
    lib.rt_version.restype = C.c_char_p
    for name in EXPORTS:
        if name not in ("rt_destroy", "rt_version"):
            getattr(lib, name).restype = C.c_int
    _lib = lib
    return lib


class RtError(RuntimeError):
    def __init__(self, fn, rc):
        super().__init__(f"{fn} failed with status {rc}")
        self.rc = rc


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class RayTracer:
    def __init__(self, spheres: np.ndarray, dim: int):
        self.lib = load_library()
        self.spheres = np.ascontiguousarray(spheres, dtype=SPHERE_DTYPE)
        self.n, self.dim = self.spheres.shape[0], dim
        self._ctx = C.c_void_p()
        rc = self.lib.rt_create(C.byref(self._ctx), _ptr(self.spheres), self.n, dim)
        if rc != RT_OK:
            self._ctx = C.c_void_p()
            raise RtError("rt_create", rc)

    def close(self):
        if getattr(self, "_ctx", None) and self._ctx.value:
            self.lib.rt_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_mode(self, mode: int):
        rc = self.lib.rt_set_mode(self._ctx, mode)
        if rc:
            raise RtError("rt_set_mode", rc)

    # ---- device-resident animation state (sphere.cuh:50-118)
    def _chk(self, fn, rc):
        if rc:
            raise RtError(fn, rc)

    def anim_init(self):
        self._chk("rt_anim_init", self.lib.rt_anim_init(self._ctx))

    def anim_axis_move(self, shake_width=35):
        self._chk("rt_anim_axis_move", self.lib.rt_anim_axis_move(self._ctx, shake_width))

    def anim_curve_move(self):
        self._chk("rt_anim_curve_move", self.lib.rt_anim_curve_move(self._ctx))

    def anim_update_speed_angle(self, update_prob=1, max_speed=18):
        self._chk("rt_anim_update_speed_angle", self.lib.rt_anim_update_speed_angle(self._ctx, update_prob, max_speed))

    def anim_state(self):
        sh = np.zeros((self.n, 4), dtype=np.int32); ang = np.zeros(self.n, dtype=np.float64); rng = np.zeros((self.n, 6), dtype=np.uint32)
        self._chk("rt_anim_get_state", self.lib.rt_anim_get_state(self._ctx, _ptr(sh), _ptr(ang), _ptr(rng)))
        return sh, ang, rng

    def anim_loop(self, frames, shake=2, shake_width=35, update_prob=1, max_speed=18, c_shift_x=0, c_shift_y=0, download=True):
        """generate_frame `frames` times on the device (rt_anim_loop); returns the last frame (or None); stats().ms_render = device time per frame."""
        img = np.zeros((self.dim, self.dim, 4), dtype=np.uint8) if download else None
        self._chk("rt_anim_loop", self.lib.rt_anim_loop(self._ctx, frames, shake, shake_width, update_prob, max_speed, c_shift_x, c_shift_y, _ptr(img)))
        return img

    def render_repeat(self, shifts, frames, c_shift_x=0, c_shift_y=0, download=True):
        """The same frame `frames` times back to back; stats().ms_render is then the device time per frame."""
        sh = np.ascontiguousarray(shifts, dtype=np.int32).reshape(-1, 4)
        img = np.zeros((self.dim, self.dim, 4), dtype=np.uint8) if download else None
        rc = self.lib.rt_render_repeat(self._ctx, _ptr(sh), c_shift_x, c_shift_y, frames, _ptr(img))
        if rc:
            raise RtError("rt_render_repeat", rc)
        return img

    def render(self, shifts, c_shift_x=0, c_shift_y=0, rows=None, download=True):
        """shifts=None: render from the device-resident animation state (after anim_init)."""
        sh = None
        if shifts is not None:
            sh = np.ascontiguousarray(shifts, dtype=np.int32).reshape(-1, 4)
            assert sh.shape[0] == self.n
        y0, y1 = (0, self.dim) if rows is None else rows
        img = np.zeros((y1 - y0, self.dim, 4), dtype=np.uint8) if download else None
        rc = self.lib.rt_render_rows(self._ctx, _ptr(sh), c_shift_x, c_shift_y, y0, y1, _ptr(img))
        if rc:
            raise RtError("rt_render_rows", rc)
        return img

    def stats(self) -> RtStats:
        s = RtStats()
        rc = self.lib.rt_get_stats(self._ctx, C.byref(s))
        if rc:
            raise RtError("rt_get_stats", rc)
        return s


def init_shifts(n: int):
    sh = np.zeros((n, 4), dtype=np.int32)
    ang = np.zeros(n, dtype=np.float64)
    rc = load_library().rt_init_shifts(n, _ptr(sh), _ptr(ang))
    if rc:
        raise RtError("rt_init_shifts", rc)
    return sh, ang


def version() -> str:
    return load_library().rt_version().decode()
