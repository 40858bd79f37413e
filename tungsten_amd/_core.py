"""ctypes binding of libtungsten_amd.so (HIP core, C ABI of include/tungsten_amd.h)."""
import ctypes as C
import os

import numpy as np

from . import _abi

_CORE_LIB_PATH = os.environ.get("TGPU_CORE_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libtungsten_amd.so")  # env: A/B builds
_lib = None


class CoreError(RuntimeError):
    pass


TgpuStats = _abi.Stats


def load_core(path):
    """Load one build of the HIP core from `path` and declare its C ABI (A/B benchmarking loads two)."""
    if True:
        if not os.path.exists(path):
            raise CoreError(f"{path} is missing: the HIP extension has not been built "
                            "(python -c 'import __graft_entry__ as g; g.build()'). There is no CPU fallback.")
        try:
            import torch  # noqa: F401
        except Exception:
            pass
        L = C.CDLL(path)
        vp = C.c_void_p
        L.tgpu_init.argtypes = [C.c_int, C.POINTER(vp)]
        L.tgpu_destroy.argtypes = [vp]
        L.tgpu_last_error.restype = C.c_char_p
        L.tgpu_last_error.argtypes = [vp]
        L.tgpu_upload_scene.argtypes = [vp, vp, vp, C.c_uint64]
        L.tgpu_render_pass.argtypes = [vp, C.c_uint32, C.c_uint32, vp]
        L.tgpu_wait.argtypes = [vp]
        L.tgpu_abort.argtypes = [vp]
        L.tgpu_download.argtypes = [vp, vp, vp]
        L.tgpu_resolve_to_device.argtypes = [vp, vp, vp]
        L.tgpu_upload_framebuffer.argtypes = [vp, vp, vp]
        L.tgpu_trace_closest.argtypes = [vp, vp, C.c_uint32, vp]
        L.tgpu_trace_occluded.argtypes = [vp, vp, C.c_uint32, vp]
        L.tgpu_set_profiling.argtypes = [vp, C.c_uint32]
        L.tgpu_get_stats.argtypes = [vp, C.POINTER(_abi.Stats)]
        L.tgpu_reset_stats.argtypes = [vp]
        L.tgpu_eval_math.argtypes = [vp, C.c_int, vp, vp, C.c_uint32, vp]
        L.tgpu_render_pass_adaptive.argtypes = [vp, vp, vp, vp]
        L.tgpu_download_records.argtypes = [vp, vp]
        L.tgpu_upload_records.argtypes = [vp, vp]
        L.tgpu_download_output_buffer.argtypes = [vp, C.c_int, C.c_int, vp]
        L.tgpu_upload_output_buffer.argtypes = [vp, C.c_int, C.c_int, vp]
    return L


def core_lib():
    """The in-tree HIP core.  torch (when installed) is imported first so that both share ONE HIP
    runtime (torch bundles libamdhip64.so.7; the loader then reuses it for our NEEDED entry)."""
    global _lib
    if _lib is None:
        _lib = load_core(os.environ.get("TGPU_CORE_LIB", _CORE_LIB_PATH))  # TGPU_CORE_LIB: another build of the same core (A/B runs)
    return _lib


EXPORTED_SYMBOLS = [
    "tgpu_init", "tgpu_destroy", "tgpu_last_error", "tgpu_upload_scene", "tgpu_render_pass", "tgpu_wait",
    "tgpu_abort", "tgpu_download", "tgpu_resolve_to_device", "tgpu_upload_framebuffer", "tgpu_trace_closest",
    "tgpu_trace_occluded", "tgpu_set_profiling", "tgpu_get_stats", "tgpu_reset_stats", "tgpu_eval_math",
    "tgpu_render_pass_adaptive", "tgpu_download_records", "tgpu_upload_records", "tgpu_download_output_buffer",
    "tgpu_upload_output_buffer",
]


class Context:
    """One GPU, one scene, one image-plane shard (the Integrator-side view of the C ABI)."""

    def __init__(self, device=0, lib=None):
        L = self._lib = lib if lib is not None else core_lib()
        h = C.c_void_p()
        rc = L.tgpu_init(int(device), C.byref(h))
        if rc != 0 or not h:
            raise CoreError(f"tgpu_init(device={device}) failed with {rc}: no usable HIP device. "
                            "This library has no CPU fallback.")
        self._h = h
        self._scene = self._shard = None

    def _check(self, rc):
        if rc != 0:
            msg = self._lib.tgpu_last_error(self._h)
            raise CoreError(f"tgpu error {rc}: {msg.decode() if msg else ''}")

    def upload(self, host_scene, shard, max_paths=0):
        self._scene, self._shard = host_scene, shard
        self._check(self._lib.tgpu_upload_scene(self._h, C.cast(host_scene.desc, C.c_void_p),
                                                 C.cast(shard.desc, C.c_void_p), int(max_paths)))

    def render_pass(self, spp_begin, spp_count, stream=None):
        self._check(self._lib.tgpu_render_pass(self._h, spp_begin, spp_count, C.c_void_p(stream or 0)))

    def render_pass_adaptive(self, vt_sample_index, vt_sample_count, stream=None):
        """One pass with per-variance-tile sample counts (whole-image tables, varW*varH entries)."""
        idx = np.ascontiguousarray(vt_sample_index, np.uint32)
        cnt = np.ascontiguousarray(vt_sample_count, np.uint32)
        n = ((self._shard.res_x + 3)//4)*((self._shard.res_y + 3)//4)
        if idx.size != n or cnt.size != n:
            raise ValueError(f"expected {n} variance tiles, got {idx.size}/{cnt.size}")
        self._check(self._lib.tgpu_render_pass_adaptive(self._h, idx.ctypes.data, cnt.ctypes.data, C.c_void_p(stream or 0)))

    def download_records(self, table=None):
        """SampleRecord statistics of this shard's variance tiles, written into the whole-image table."""
        from ._host import RECORD_DTYPE
        n = ((self._shard.res_x + 3)//4)*((self._shard.res_y + 3)//4)
        table = np.zeros(n, RECORD_DTYPE) if table is None else table
        assert table.dtype == RECORD_DTYPE and table.size == n and table.flags.c_contiguous
        self._check(self._lib.tgpu_download_records(self._h, table.ctypes.data))
        return table

    def upload_records(self, table):
        from ._host import RECORD_DTYPE
        table = np.ascontiguousarray(table, RECORD_DTYPE)
        self._check(self._lib.tgpu_upload_records(self._h, table.ctypes.data))

    def wait(self):
        self._check(self._lib.tgpu_wait(self._h))

    def render(self, spp, spp_step=16, spp_begin=0):
        done = spp_begin
        while done < spp_begin + spp:
            k = min(spp_step, spp_begin + spp - done)
            self.render_pass(done, k)
            done += k
        self.wait()

    def download(self):
        n = self._shard.n_pixels
        rgb = np.zeros((n, 3), np.float32)
        cnt = np.zeros(n, np.uint32)
        self._check(self._lib.tgpu_download(self._h, rgb.ctypes.data, cnt.ctypes.data))
        return rgb, cnt

    def download_output_buffer(self, name, part="a"):
        """One part ("a", "b", "variance", "count") of an output buffer ("color", "depth", "normal", "albedo", "visibility") of the
        shard: (n_pixels, channels) float32, or (n_pixels,) uint32 counts.  Raises for a buffer / part not requested at upload."""
        t = _abi.OUTPUT_NAMES.index(name)
        p = ("a", "b", "variance", "count").index(part)
        n = self._shard.n_pixels
        out = np.zeros(n, np.uint32) if p == _abi.PART_COUNT else np.zeros((n, _abi.OUTPUT_CHANNELS[t]), np.float32)
        self._check(self._lib.tgpu_download_output_buffer(self._h, t, p, out.ctypes.data))
        return out

    def upload_output_buffer(self, name, part, values):
        t = _abi.OUTPUT_NAMES.index(name)
        p = ("a", "b", "variance", "count").index(part)
        values = np.ascontiguousarray(values, np.uint32 if p == _abi.PART_COUNT else np.float32)
        self._check(self._lib.tgpu_upload_output_buffer(self._h, t, p, values.ctypes.data))

    def resolve_to_device(self, device_ptr, stream=None):
        self._check(self._lib.tgpu_resolve_to_device(self._h, C.c_void_p(device_ptr), C.c_void_p(stream or 0)))

    def trace_closest(self, rays):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        hits = np.zeros((len(rays), 4), np.float32)
        self._check(self._lib.tgpu_trace_closest(self._h, rays.ctypes.data, len(rays), hits.ctypes.data))
        return hits

    def trace_occluded(self, rays):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        occ = np.zeros(len(rays), np.uint8)
        self._check(self._lib.tgpu_trace_occluded(self._h, rays.ctypes.data, len(rays), occ.ctypes.data))
        return occ

    def eval_math(self, kind, a, b=None):
        a = np.ascontiguousarray(a, np.float32)
        b = np.ascontiguousarray(b if b is not None else np.zeros_like(a), np.float32)
        out = np.zeros_like(a)
        self._check(self._lib.tgpu_eval_math(self._h, kind, a.ctypes.data, b.ctypes.data, a.size, out.ctypes.data))
        return out

    def set_profiling(self, flags):
        self._check(self._lib.tgpu_set_profiling(self._h, flags))

    def stats(self):
        s = _abi.Stats()
        self._check(self._lib.tgpu_get_stats(self._h, C.byref(s)))
        return {k: getattr(s, k) for k, _ in s._fields_}

    def reset_stats(self):
        self._check(self._lib.tgpu_reset_stats(self._h))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.tgpu_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
