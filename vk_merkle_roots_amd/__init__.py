"""MI355X-native Merkle-root engine (SHA-256d map + pairwise reduce) -- Python harness.

The product is the C ABI in include/vkmr_hip.h (HIP kernels for gfx950) and the C++
front end under csrc/host.  This package only holds what tests and bench.py need to
drive that ABI from Python: the ctypes stub (`_abi`), a small RAII-style wrapper over
device buffers (`engine`) and the build recipes (`build`).  Nothing here computes a
hash on the CPU; without the built HIP extension every entry point raises.
"""
from . import build  # noqa: F401
from ._abi import Digest, Metadata, VkmrError, check, host_lib, lib  # noqa: F401
from .engine import HipDevice, PackedBatch, RndmStream, merkle_root_packed, merkle_root_packed_batched, pack_lines, rndm_packed, tree_height  # noqa: F401
