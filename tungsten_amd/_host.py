"""ctypes binding of libtungsten_host.so (scene loading / flattening / shards / image I/O)."""
import ctypes as C
import os

import numpy as np

from . import _abi

REPO_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_HOST_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libtungsten_host.so")
_lib = None


def host_lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_HOST_LIB_PATH):
            raise RuntimeError(f"{_HOST_LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        os.environ.setdefault("TUNGSTEN_AMD_ASSETS", os.path.join(REPO_ROOT, "assets"))
        lib = C.CDLL(_HOST_LIB_PATH)
        lib.tgh_scene_load.restype = C.c_void_p
        lib.tgh_scene_load.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
        lib.tgh_scene_free.argtypes = [C.c_void_p]
        lib.tgh_scene_set_resolution.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        lib.tgh_scene_set_bounces.argtypes = [C.c_void_p, C.c_int, C.c_int]
        lib.tgh_scene_set_light_sampling.argtypes = [C.c_void_p, C.c_int]
        lib.tgh_scene_set_sobol.argtypes = [C.c_void_p, C.c_int]
        lib.tgh_scene_add_output_buffer.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        lib.tgh_scene_flatten.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        lib.tgh_scene_desc.restype = C.POINTER(_abi.SceneDesc)
        lib.tgh_scene_desc.argtypes = [C.c_void_p]
        lib.tgh_scene_info.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
        lib.tgh_scene_tonemap.restype = C.c_char_p
        lib.tgh_scene_tonemap.argtypes = [C.c_void_p]
        lib.tgh_shard_create.restype = C.c_void_p
        lib.tgh_shard_create.argtypes = [C.c_uint32]*5
        lib.tgh_shard_free.argtypes = [C.c_void_p]
        lib.tgh_shard_desc.restype = C.POINTER(_abi.ShardDesc)
        lib.tgh_shard_desc.argtypes = [C.c_void_p]
        lib.tgh_shard_size.restype = C.c_uint32
        lib.tgh_shard_size.argtypes = [C.c_void_p]
        lib.tgh_adaptive_create.restype = C.c_void_p
        lib.tgh_adaptive_create.argtypes = [C.c_uint32]*3
        lib.tgh_adaptive_free.argtypes = [C.c_void_p]
        lib.tgh_adaptive_tiles.restype = C.c_uint32
        lib.tgh_adaptive_tiles.argtypes = [C.c_void_p]
        lib.tgh_adaptive_needs_statistics.argtypes = [C.c_int, C.c_uint32]
        lib.tgh_adaptive_set_statistics.argtypes = [C.c_void_p, C.c_void_p]
        lib.tgh_adaptive_generate.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]
        lib.tgh_save_pfm.argtypes = [C.c_char_p, C.POINTER(C.c_float), C.c_int, C.c_int]
        lib.tgh_save_png_tonemapped.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_float), C.c_int, C.c_int]
        _lib = lib
    return _lib


class HostScene:
    """Scene::load + makeTraceable bakes (reference: src/core/io/Scene.cpp:373-391)."""

    def __init__(self, json_path, resolution=None, max_bounces=None, light_sampling=None, sobol=None, output_buffers=None):
        """output_buffers: extra entries of the renderer's "output_buffers" list, each a name ("color", "depth", "normal", "albedo",
        "visibility") or a tuple (name, two_buffer_variance, sample_variance)."""
        lib = host_lib()
        err = C.create_string_buffer(1024)
        self._h = lib.tgh_scene_load(os.fspath(json_path).encode(), err, len(err))
        if not self._h:
            raise RuntimeError(err.value.decode())
        if resolution is not None:
            lib.tgh_scene_set_resolution(self._h, int(resolution[0]), int(resolution[1]))
        if max_bounces is not None:
            lib.tgh_scene_set_bounces(self._h, 0, int(max_bounces))
        if light_sampling is not None:
            lib.tgh_scene_set_light_sampling(self._h, int(bool(light_sampling)))
        if sobol is not None:
            lib.tgh_scene_set_sobol(self._h, int(bool(sobol)))
        for ob in output_buffers or ():
            name, two, var = (ob, False, False) if isinstance(ob, str) else ob
            lib.tgh_scene_add_output_buffer(self._h, _abi.OUTPUT_NAMES.index(name), int(bool(two)), int(bool(var)))
        if lib.tgh_scene_flatten(self._h, err, len(err)) != 0:
            msg = err.value.decode()
            lib.tgh_scene_free(self._h)
            self._h = None
            raise RuntimeError(msg)
        info = (C.c_uint32*8)()
        lib.tgh_scene_info(self._h, info)
        self.res_x, self.res_y, self.spp, self.spp_step = info[0], info[1], info[2], info[3]
        self.use_sobol, self.adaptive = bool(info[4]), bool(info[5])
        self.min_bounces, self.max_bounces = int(info[6]), int(info[7])
        self.tonemap = lib.tgh_scene_tonemap(self._h).decode()

    @property
    def desc(self):
        """POINTER(tgpu_scene_desc); valid while this object lives."""
        return host_lib().tgh_scene_desc(self._h)

    def close(self):
        if getattr(self, "_h", None):
            host_lib().tgh_scene_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


RECORD_DTYPE = np.dtype([("sample_count", np.uint32), ("mean", np.float32), ("running_variance", np.float32)])


class AdaptiveScheduler:
    """generateWork of PathTraceIntegrator (PathTraceIntegrator.cpp:44-134) over the whole image's 4x4
    variance tiles; the C++ class of host/adaptive.hpp, the same one the C++ Integrator mirror uses."""

    def __init__(self, res_x, res_y, seed=0xBA5EBA11):
        self._h = host_lib().tgh_adaptive_create(res_x, res_y, seed & 0xFFFFFFFF)
        self.n_tiles = host_lib().tgh_adaptive_tiles(self._h)
        self.var_w, self.var_h = (res_x + 3)//4, (res_y + 3)//4

    @staticmethod
    def needs_statistics(adaptive, current_spp):
        return bool(host_lib().tgh_adaptive_needs_statistics(int(adaptive), current_spp))

    def set_statistics(self, table):
        table = np.ascontiguousarray(table, RECORD_DTYPE)
        assert table.size == self.n_tiles
        host_lib().tgh_adaptive_set_statistics(self._h, table.ctypes.data)

    def generate(self, current_spp, next_spp, adaptive=True):
        """(sample_index, sample_count) per variance tile for the pass, or None when converged."""
        idx = np.zeros(self.n_tiles, np.uint32)
        cnt = np.zeros(self.n_tiles, np.uint32)
        ok = host_lib().tgh_adaptive_generate(self._h, current_spp, next_spp, int(adaptive), idx.ctypes.data, cnt.ctypes.data)
        return (idx, cnt) if ok else None

    def __del__(self):
        try:
            if self._h:
                host_lib().tgh_adaptive_free(self._h)
                self._h = None
        except Exception:
            pass


class Shard:
    """Image-plane shard of one rank: 16x16 tiles seeded as PathTraceIntegrator::diceTiles does."""

    def __init__(self, res_x, res_y, seed=0xBA5EBA11, rank=0, world=1):
        lib = host_lib()
        self._h = lib.tgh_shard_create(res_x, res_y, seed & 0xFFFFFFFF, rank, world)
        self.res_x, self.res_y = res_x, res_y
        self.n_pixels = lib.tgh_shard_size(self._h)

    @property
    def desc(self):
        return host_lib().tgh_shard_desc(self._h)

    @property
    def pixel_index(self):
        if self.n_pixels == 0:
            return np.zeros(0, np.uint32)
        d = self.desc.contents
        return np.ctypeslib.as_array(d.pixel_index, shape=(self.n_pixels,)).copy()

    def scatter(self, rgb_shard, image=None):
        """Place shard-ordered rgb (n_pixels,3) into a (res_y,res_x,3) image."""
        if image is None:
            image = np.zeros((self.res_y*self.res_x, 3), np.float32)
        else:
            image = image.reshape(-1, 3)
        image[self.pixel_index] = rgb_shard.reshape(-1, 3)
        return image.reshape(self.res_y, self.res_x, 3)

    def close(self):
        if getattr(self, "_h", None):
            host_lib().tgh_shard_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def save_pfm(path, rgb):
    rgb = np.ascontiguousarray(rgb, np.float32)
    h, w = rgb.shape[:2]
    if host_lib().tgh_save_pfm(os.fspath(path).encode(), rgb.ctypes.data_as(C.POINTER(C.c_float)), w, h) != 0:
        raise IOError(f"cannot write {path}")


def load_pfm(path):
    """Read a colour PFM as written by the reference (ImageIO.cpp:528-545): bottom-to-top rows, LE."""
    with open(path, "rb") as f:
        magic = f.readline().strip()
        if magic != b"PF":
            raise IOError("not a colour PFM")
        w, h = (int(t) for t in f.readline().split())
        scale = float(f.readline())
        data = np.frombuffer(f.read(w*h*12), dtype="<f4" if scale < 0 else ">f4")
    return data.reshape(h, w, 3)[::-1].astype(np.float32).copy()
