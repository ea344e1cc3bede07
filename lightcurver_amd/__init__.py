"""MI355X-native implementation of lightcurver's PSF-fit + joint forward-model hot path.

The compute lives in ``csrc/`` (hand-written HIP for gfx950 behind the C ABI declared in
``include/lcmi.h``); the Python modules mirror the STARRED call sites of the reference
(``starred_api``) and its step functions (``processes``).  There is no CPU fallback: every
compute entry point raises if ``liblcmi.so`` is missing or no GPU is visible.
"""
__version__ = '0.1.0'
