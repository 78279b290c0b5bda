"""Host-side binding of the MI355X denoising kernels.

``Engine`` wraps one ``adn_ctx`` (one GPU, one stream).  PyTorch-ROCm tensors are
used purely as device buffers: every computation happens in hand-written HIP
kernels reached through the C ABI in ``include/adn.h``.
"""
from ._capi import AdnError, LIB_PATH, exported_symbols  # noqa: F401
from .engine import Engine, get_engine  # noqa: F401
