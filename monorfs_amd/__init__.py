"""monorfs_amd — MI355X (gfx950) implementation of monorfs's RB-PHD-SLAM inner loop.

The product is the C-ABI shared library `monorfs_amd/csrc/libphdhip.so` (include/phdhip.h);
this package is the thin Python host side used by the tests and the benchmark."""
from .abi import *  # noqa: F401,F403
