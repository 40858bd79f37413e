"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
Nothing under tungsten_amd/ may import this module.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None


class OracleStats(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("extend_rays", C.c_uint64), ("shadow_rays", C.c_uint64),
                ("surface_events", C.c_uint64), ("node_visits", C.c_uint64), ("tri_tests", C.c_uint64)]


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(f"{_LIB_PATH} missing: run `make oracle` (or __graft_entry__.build())")
        L = C.CDLL(_LIB_PATH)
        L.oracle_create.restype = C.c_void_p
        L.oracle_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.oracle_destroy.argtypes = [C.c_void_p]
        L.oracle_render_image.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_int]
        L.oracle_render_shard.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_int]
        L.oracle_render_reference.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_int]
        L.oracle_render_shard_adaptive.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                   C.c_void_p, C.c_int]
        L.oracle_exp_table.argtypes = [C.c_void_p, C.c_void_p]
        L.oracle_fast_exp.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_trace_closest.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        L.oracle_trace_occluded.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        L.oracle_hash32.restype = C.c_uint32
        L.oracle_hash32.argtypes = [C.c_uint32]
        L.oracle_sobol.restype = C.c_uint32
        L.oracle_sobol.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
        L.oracle_pcg_stream.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p]
        L.oracle_camera_ray.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_float, C.c_float, C.c_void_p]
        L.oracle_tile_seeds.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        L.oracle_set_brute_force.argtypes = [C.c_void_p, C.c_int]
        L.oracle_bind_output_buffers.argtypes = [C.c_void_p, C.c_void_p]
        L.oracle_kat_output_buffer.argtypes = [C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_kat_embree_user.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_kat_embree_tris.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        L.oracle_get_stats.argtypes = [C.c_void_p, C.POINTER(OracleStats)]
        L.oracle_reset_stats.argtypes = [C.c_void_p]
        _lib = L
    return _lib


REF = dict(rng=0, math=0, rcp=0)        # the reference's behaviour (tile-sequential PCG, glibc libm)
REF_RCP = dict(rng=0, math=0, rcp=1)    # + Embree's RCPPS-based rcp() (x86 only): triangle t,u,v and the node test that decides coplanar quad/cube ties
GPU = dict(rng=1, math=1, rcp=0)        # what the HIP core computes (path-local PCG, tgpu_math.h)


class Oracle:
    def __init__(self, host_scene, rng=1, math=1, rcp=0):
        self._scene = host_scene  # keeps the flattened buffers alive
        self._h = lib().oracle_create(C.cast(host_scene.desc, C.c_void_p), rng, math, rcp)
        if not self._h:
            raise RuntimeError("oracle_create failed (ABI version mismatch?)")
        self.res_x, self.res_y = host_scene.res_x, host_scene.res_y

    def render_image(self, spp, seed=0xBA5EBA11, spp_step=16, threads=None):
        """Whole image in the reference's tile order; returns (H,W,3) mean and (H,W) counts."""
        threads = threads or os.cpu_count() or 1
        n = self.res_x*self.res_y
        rgb = np.zeros((n, 3), np.float32)
        cnt = np.zeros(n, np.uint32)
        done = 0
        while done < spp:
            k = min(spp_step, spp - done)
            lib().oracle_render_image(self._h, seed & 0xFFFFFFFF, done, k, rgb.ctypes.data, cnt.ctypes.data, threads)
            done += k
        return rgb.reshape(self.res_y, self.res_x, 3), cnt.reshape(self.res_y, self.res_x)

    def render_reference(self, spp, seed=0xBA5EBA11, spp_step=16, adaptive=True, threads=None):
        """The whole StandaloneRenderer pass loop incl. generateWork / adaptive sampling
        (PathTraceIntegrator.cpp:44-156); returns (H,W,3) mean, (H,W) counts and the (varH,varW,3)
        {sampleCount, mean, runningVariance} SampleRecord table."""
        threads = threads or os.cpu_count() or 1
        n = self.res_x*self.res_y
        var_w, var_h = (self.res_x + 3)//4, (self.res_y + 3)//4
        rgb = np.zeros((n, 3), np.float32)
        cnt = np.zeros(n, np.uint32)
        rec = np.zeros((var_h, var_w, 3), np.float32)
        lib().oracle_render_reference(self._h, seed & 0xFFFFFFFF, spp, spp_step, int(adaptive), rgb.ctypes.data,
                                      cnt.ctypes.data, rec.ctypes.data, threads)
        return rgb.reshape(self.res_y, self.res_x, 3), cnt.reshape(self.res_y, self.res_x), rec

    def render_shard(self, shard, spp, spp_step=16, threads=None, rgb=None, cnt=None, spp_begin=0):
        threads = threads or os.cpu_count() or 1
        n = shard.n_pixels
        rgb = np.zeros((n, 3), np.float32) if rgb is None else rgb
        cnt = np.zeros(n, np.uint32) if cnt is None else cnt
        done = spp_begin
        while done < spp_begin + spp:
            k = min(spp_step, spp_begin + spp - done)
            rc = lib().oracle_render_shard(self._h, C.cast(shard.desc, C.c_void_p), done, k, rgb.ctypes.data,
                                           cnt.ctypes.data, threads)
            if rc != 0:
                raise RuntimeError("oracle_render_shard needs rng mode 1")
            done += k
        return rgb, cnt

    def render_shard_adaptive(self, shard, vt_index, vt_count, rgb, cnt, records, threads=None):
        """One pass with per-variance-tile counts; rgb/cnt (shard order) and the whole-image record table
        (numpy, tungsten_amd.RECORD_DTYPE) are updated in place -- what tgpu_render_pass_adaptive does."""
        threads = threads or os.cpu_count() or 1
        vt_index = np.ascontiguousarray(vt_index, np.uint32)
        vt_count = np.ascontiguousarray(vt_count, np.uint32)
        assert records.flags.c_contiguous and records.itemsize == 12
        rc = lib().oracle_render_shard_adaptive(self._h, C.cast(shard.desc, C.c_void_p), vt_index.ctypes.data,
                                                vt_count.ctypes.data, rgb.ctypes.data, cnt.ctypes.data,
                                                records.ctypes.data, threads)
        if rc != 0:
            raise RuntimeError("oracle_render_shard_adaptive needs rng mode 1")
        return rgb, cnt, records

    def bind_output_buffers(self, n_pixels):
        """Allocates the output buffers the scene's renderer settings request (tgpu_scene_desc.output_buffers: the colour buffer with
        its optional parts, and the auxiliary depth / normal / albedo / visibility buffers) for `n_pixels` pixels in the order of the
        render call that follows (shard order for render_shard*, image order for render_image), binds them, and returns
        {name: {"a", "b", "variance", "count"}} with numpy arrays (absent parts are left out)."""
        from tungsten_amd import _abi

        class OutBuf(C.Structure):
            _fields_ = [("a", C.c_void_p), ("b", C.c_void_p), ("variance", C.c_void_p), ("count", C.c_void_p)]

        settings = self._scene.desc.contents.output_buffers
        table = (OutBuf*_abi.OUTPUT_COUNT)()
        out = {}
        for t, name in enumerate(_abi.OUTPUT_NAMES):
            ob = settings[t]
            if t != _abi.OUTPUT_COLOR and not ob.enabled:
                continue
            ch = _abi.OUTPUT_CHANNELS[t]
            parts = {"a": np.zeros((n_pixels, ch), np.float32), "count": np.zeros(n_pixels, np.uint32)}
            if ob.two_buffer_variance:
                parts["b"] = np.zeros((n_pixels, ch), np.float32)
            if ob.sample_variance:
                parts["variance"] = np.zeros((n_pixels, ch), np.float32)
            for key, arr in parts.items():
                setattr(table[t], key, arr.ctypes.data)
            out[name] = parts
        self._outputs = (table, out)  # keeps the arrays alive while they are bound
        lib().oracle_bind_output_buffers(self._h, C.cast(table, C.c_void_p))
        return out

    def set_brute_force(self, on=True):
        """Triangle queries test every triangle (no BVH): the definition both BVH walks must reproduce."""
        lib().oracle_set_brute_force(self._h, int(on))

    def trace_closest(self, rays):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        hits = np.zeros((len(rays), 4), np.float32)
        lib().oracle_trace_closest(self._h, rays.ctypes.data, len(rays), hits.ctypes.data)
        return hits

    def trace_occluded(self, rays):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        occ = np.zeros(len(rays), np.uint8)
        lib().oracle_trace_occluded(self._h, rays.ctypes.data, len(rays), occ.ctypes.data)
        return occ

    def camera_ray(self, px, py, xi0, xi1):
        out = np.zeros(6, np.float32)
        lib().oracle_camera_ray(self._h, px, py, xi0, xi1, out.ctypes.data)
        return out

    def sobol(self, index, dim, scramble=0):
        return lib().oracle_sobol(self._h, index, dim, scramble)

    def stats(self):
        s = OracleStats()
        lib().oracle_get_stats(self._h, C.byref(s))
        return {k: getattr(s, k) for k, _ in s._fields_}

    def reset_stats(self):
        lib().oracle_reset_stats(self._h)

    def close(self):
        if getattr(self, "_h", None):
            lib().oracle_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def tile_seeds(res_x, res_y, seed=0xBA5EBA11):
    n = ((res_x + 15)//16)*((res_y + 15)//16)
    out = np.zeros(n, np.uint32)
    lib().oracle_tile_seeds(res_x, res_y, seed & 0xFFFFFFFF, out.ctypes.data)
    return out


def pcg_stream(state, seq, n):
    out = np.zeros(n, np.uint32)
    lib().oracle_pcg_stream(state, seq, n, out.ctypes.data)
    return out
