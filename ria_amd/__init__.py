"""ria_amd — MI355X-native RX signal chain for the RIA HF modem (hot path only)."""
from . import capi  # noqa: F401
from .build import build  # noqa: F401

__all__ = ["capi", "build", "RxEngine"]


def __getattr__(name):
    if name == "RxEngine":
        from .engine import RxEngine
        return RxEngine
    raise AttributeError(name)
