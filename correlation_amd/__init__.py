"""correlation_amd - MI355X (gfx950) Lucas-Kanade image-correlation engine.

The product is the C-ABI library `liblk_engine.so` (include/lk_engine.h) built from
correlation_amd/csrc/ by `python -m correlation_amd.build`.  This package is the thin
Python host used by the tests and the benchmark; it holds no compute path of its own
and fails loudly if the library is missing.
"""
from ._ffi import (ERROR_BAD_DOMAIN, ERROR_CORRELATION_MAX_ITERS_REACHED, ERROR_DEVICE,  # noqa: F401
                   ERROR_INTERPOLATION_OUT_OF_IMAGE, ERROR_NONE, FM_U, FM_UV, FM_UVQ,
                   FM_UVUXUYVXVY, IM_BICUBIC, IM_BICUBIC_SEPARABLE, IM_BILINEAR, IM_NEAREST, IMG_DEF, IMG_NXT, IMG_UND,
                   LIB_PATH, N_PARAMS, RESULT_DTYPE, SYMBOLS, load_library)
from .engine import HipCorrelationEngine, LkError  # noqa: F401
from . import speckle  # noqa: F401
from . import tracker  # noqa: F401
from .group import HipCorrelationGroup  # noqa: F401
