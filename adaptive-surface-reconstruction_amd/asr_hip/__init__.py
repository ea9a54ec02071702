"""MI355X-native hot path of adaptive surface reconstruction (host-side package).

`libasr_hip.so` (csrc/, C ABI in include/asr_hip.h) does all the work; this package binds it
with ctypes and uses torch only for device memory and streams.
"""
from ._lib import AsrHipError, Context, LIB_PATH, load  # noqa: F401
