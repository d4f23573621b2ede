"""magnet_amd — MI355X-native implementation of MagNet's complex magnetic-Laplacian
propagation (Laplacian builder, fused complex Chebyshev SpMM, ChebNet / ChebNet_Edge),
drop-in for the reference's `utils.hermitian` and `layer.sparse_magnet` modules.

The compute path is libmagnet_hip.so (hand-written HIP for gfx950, C ABI in
include/magnet_hip.h); importing this package without the built library fails.
"""
from ._lib import MagnetHipError                                   # noqa: F401  (fails loudly when not built)
from .laplacian import MagneticLaplacian, build_magnetic_laplacian, pack_complex  # noqa: F401

__all__ = ['MagneticLaplacian', 'build_magnetic_laplacian', 'pack_complex', 'MagnetHipError']
