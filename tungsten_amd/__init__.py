"""tungsten_amd -- MI355X wavefront path-tracing core for Tungsten's ``path_tracer``.

Python is only the harness language here (tests, bench, multi-GPU launch); the
product is ``libtungsten_amd.so`` (HIP kernels behind the C ABI declared in
``include/tungsten_amd.h``) plus the C++ host library ``libtungsten_host.so``.
Both are bound with ctypes.  There is no CPU fallback: creating a
:class:`Context` without the HIP library or without a GPU raises.
"""
from ._host import HostScene, Shard, AdaptiveScheduler, RECORD_DTYPE, host_lib, save_pfm, load_pfm, REPO_ROOT  # noqa: F401
from ._core import Context, core_lib, CoreError, TgpuStats  # noqa: F401

__all__ = ["HostScene", "Shard", "AdaptiveScheduler", "RECORD_DTYPE", "Context", "CoreError", "TgpuStats", "host_lib", "core_lib",
           "save_pfm", "load_pfm", "REPO_ROOT"]
