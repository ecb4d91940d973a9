"""`import rela` — the reference's pybind module name (rela/pybind.cc:16-93; imported by pyhanabi/create.py:16,20,
selfplay.py, eval.py, utils.py) resolved to the device pipeline of this repository: put the repository root on PYTHONPATH
ahead of the reference's build/ directory and its drivers run unchanged on libhsad.so.

The classes live in hanabi_sad_amd/rela.py (ctypes over include/hsad.h).  `__file__` names the shared library that backs the
module, which is what create.py:20 asserts about it (`rela.__file__.endswith(".so")`)."""
from hanabi_sad_amd import _lib as _hsad_lib
from hanabi_sad_amd.rela import (BatchRunner, Context, FFTransition, MultiDeviceError, R2D2Actor,  # noqa: F401
                                 RNNPrioritizedReplay, RNNTransition, ThreadLoop, aggregate_priority)

__shim__ = __file__
__file__ = _hsad_lib.LIB_PATH
