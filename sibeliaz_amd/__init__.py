"""sibeliaz_amd — MI355X-native locally-collinear-block finder (drop-in for SibeliaZ-LCB's BlocksFinder hot path).

The product is the C-ABI shared library `libsibeliaz_amd.so` (HIP kernels for gfx950 + C++ host) and the
`sibeliaz-lcb` executable built from `sibeliaz_amd/csrc/`. This Python package is a thin ctypes binding used by
the tests, `bench.py` and the multi-GPU driver; names mirror the reference classes they stand for
(JunctionStorage, BlocksFinder).
"""
from .api import (LcbError, JunctionStorage, Device, GpuSet, Comm, Committer, BlocksFinder, Params, load_library, lib_path,  # noqa: F401
                  SEED_DTYPE, INSTANCE_DTYPE, BLOCK_DTYPE, Hooks)
