"""atlas_amd — MI355X-native (gfx950) implementation of the Atlas dense-retrieval hot path.

Scope: the exact-MIPS search of ``src/index.py`` (``DistributedIndex``), its
cross-rank top-k exchange, and the index-refresh epilogue, behind the reference's own Python
API so that ``src/atlas.py`` runs unchanged.  All device work goes through hand-written HIP
kernels in ``csrc/`` via the C-ABI in ``include/atlas_hip.h``; there is no CPU or eager-PyTorch
fallback: without the built library every compute entry point raises.
"""
from .index import HipDistributedIndex  # noqa: F401
from .index_io import load_or_initialize_index, save_embeddings_and_index  # noqa: F401

__all__ = ["HipDistributedIndex", "load_or_initialize_index", "save_embeddings_and_index"]
