"""webgpu-path-tracer_amd — MI355X-native path-tracing integrator behind the reference's buffer contract.

  csrc/      HIP kernels (gfx950) + the C ABI of include/ptmi.h  -> libptmi.so
  ptmi.py    ctypes binding of that C ABI (Context, Params, Stats)
  host/      Python mirror of the reference's lib/ scene classes (Scene, Camera, ObjReader, build_bvh ...)
  js/        Node host: N-API addon + WebGPU-shaped shim + the same scene classes in JavaScript
  scenes.py  the canonical configurations of BASELINE.json / SURVEY.md §8d

The directory name has a hyphen; load it with `importlib` as module `webgpu_path_tracer_amd`
(see tests/conftest.py::load_pkg, bench.py, __graft_entry__.py).
"""
from . import _build  # noqa: F401
from . import ptmi, scenes  # noqa: F401
from .ptmi import Context, Params, PtmiError, default_params, load_library  # noqa: F401
