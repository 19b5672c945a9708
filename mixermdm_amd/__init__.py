"""mixermdm_amd -- MI355X (gfx950) implementation of MixerMDM's iterative denoising hot path.

Python here is plumbing (weight loading, the YAML/config surface, the DDIM host loop and the reference's
``src/models`` operator API); every arithmetic op of the path runs in hand-written HIP kernels behind the C ABI in
``include/mmdm.h`` (``libmmdm_hip.so``, built in-tree by ``mixermdm_amd/build.py``).  There is no CPU fallback:
importing works without a GPU, calling any operator without the library or a device raises.
"""
from ._lib import lib_path, load_library, MMDMError  # noqa: F401
