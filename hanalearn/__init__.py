"""`import hanalearn` — the reference's pybind module name (cpp/pybind.cc:14-56; imported by pyhanabi/create.py:17,21 and
eval.py) resolved to the batched device environment of this repository (hanabi_sad_amd/hanalearn.py over include/hsad.h).
`__file__` names the shared library that backs the module (create.py:21 asserts `.endswith(".so")`)."""
from hanabi_sad_amd import _lib as _hsad_lib
from hanabi_sad_amd.hanalearn import HanabiEnv, HanabiThreadLoop, HanabiVecEnv  # noqa: F401

__shim__ = __file__
__file__ = _hsad_lib.LIB_PATH
