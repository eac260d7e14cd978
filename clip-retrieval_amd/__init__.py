"""MI355X-native hot path of clip-retrieval: CLIP encode + inner-product kNN behind the reference's
own seams (ClipMapper / Runner, and the faiss-Index duck type of clip_back).  See DESIGN.md.

Nothing here computes on the CPU: every product entry point goes through `lib/libclipx.so`
(HIP, gfx950) and raises `HipLibraryError` if the library or a GPU is missing.
"""

__version__ = "0.1.0"

from ._lib import HipLibraryError, ResidualStreamOverflow, load_library, library_path  # noqa: F401
