"""image_matching_amd -- MI355X (gfx950) implementation of the SIFT extract-and-match hot path of
Fricodelco/image_matching behind the reference's own `match_finder` API.

The compute lives in libimatch.so (hand-written HIP, C-ABI in include/imatch.h); this package is
the ctypes binding plus the host-side mirror of the reference's Python interface.
"""
from ._lib import AkazeParams, Context, Features, ImatchError, SiftParams, load_library  # noqa: F401
