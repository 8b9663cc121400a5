"""liblcg_amd -- MI355X-native conjugate-gradient hot path behind liblcg's solver API.

The product is the HIP shared library ``liblcg_amd/lib/liblcg_hip.so`` (C ABI:
``include/lcg_hip.h``; C++ drop-in header: ``include/lcg_dropin.hpp``).  This package is the
host-side convenience layer used by the tests and the benchmark: ctypes prototypes
(``_lib``), liblcg-shaped entry points (``api``) and the bundled-fixture readers
(``coo_io``).  Importing it does not load the GPU library; calling into ``api`` does, and
fails loudly if the library is not built or no GPU is present.
"""
from . import coo_io  # noqa: F401

__all__ = ["coo_io"]
