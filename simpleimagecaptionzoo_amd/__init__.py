"""MI355X-native caption-decoding hot path (BUTD / SCST) behind simpleImageCaptionZoo's Captioner contract.

Host code is Python on PyTorch-ROCm (device memory, streams, torch.distributed); all compute on the path is
in libicz.so (hand-written HIP for gfx950) reached through the C ABI in include/icz.h.
"""
from ._lib import IczError, lib  # noqa: F401
