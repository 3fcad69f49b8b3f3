"""MI355X-native X-LXMERT hot path (LxmertEncoder stack + masked-visual-token head, data-parallel step)."""
from .config import XLxmertConfig  # noqa: F401

__all__ = ["XLxmertConfig"]
