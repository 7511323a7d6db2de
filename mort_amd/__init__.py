"""mort_amd -- MI355X-native render path for lgleznah/mort.

host:    C scene layer (libmort_host.so), no GPU.
hip:     C-ABI over the gfx950 kernels (libmort_hip.so); fails loudly when missing.
"""
from . import structs  # noqa: F401
