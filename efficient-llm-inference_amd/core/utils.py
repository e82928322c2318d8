"""Import-path alias: the reference keeps these helpers in ``src/core/utils.py``; here they live
in :mod:`.memory`."""
from .memory import *  # noqa: F401,F403
from .memory import get_cpu_mem_mb, get_gpu_peak_mb, kv_bytes_fp, mb, reset_gpu_peak, tensor_bytes  # noqa: F401
