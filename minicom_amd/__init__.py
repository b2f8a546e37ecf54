"""minicom_amd -- MI355X-native implementation of minicom's sketch + index + overlap hot path.

The compute lives in hand-written HIP kernels (minicom_amd/csrc/*.hip -> minicom_amd/lib/libmcom_hip.so)
behind the C ABI of include/mcom.h.  This package is the thin Python host mirror used by the tests and
bench.py; PyTorch only supplies device memory, streams and torch.distributed.
There is no CPU fallback: without the HIP library or without a GPU every operation raises.
"""
from .hip import Context, McomError, lib_path, load_library, MM_DTYPE, ABI_SYMBOLS  # noqa: F401

__all__ = ["Context", "McomError", "lib_path", "load_library", "MM_DTYPE", "ABI_SYMBOLS"]
