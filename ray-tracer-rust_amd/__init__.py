"""ray-tracer-rust_amd — MI355X-native per-pixel tracer hot path of antoinedesbois/Ray-Tracer-Rust.

csrc/   HIP kernel (gfx950) + C ABI (include/rtx.h) + host-side scene preparation -> librtx.so
host/   C++ mirror of the reference's Scene/Camera/Triangle/Light API and the rtx_host CLI
rtx.py  ctypes binding of the C ABI (plumbing for tests and bench.py)

The directory name is not a Python identifier; import it with
    importlib.import_module("ray-tracer-rust_amd")
"""
from .rtx import *  # noqa: F401,F403
from . import rtx  # noqa: F401
