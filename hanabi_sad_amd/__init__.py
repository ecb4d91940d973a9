"""hanabi_sad_amd — MI355X-native hot path of facebookresearch/hanabi_SAD.

Host-side mirror of the reference's `hanalearn` / `rela` surface on top of libhsad.so
(C ABI in include/hsad.h; HIP kernels in hanabi_sad_amd/csrc).  There is no CPU fallback:
importing the compute entry points without the built library raises.
"""
from ._lib import load_library, build_library, HsadError  # noqa: F401
from .env import BatchedHanabiEnv  # noqa: F401

__all__ = ["load_library", "build_library", "HsadError", "BatchedHanabiEnv"]
