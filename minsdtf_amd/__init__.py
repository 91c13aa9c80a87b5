"""minsdtf_amd — MI355X (gfx950) native SD1.5 denoise hot path behind minSDTF's StableDiffusion API.

Public surface mirrors the reference package (stable_diffusion/__init__.py:14-19) for the
components on the accelerated path."""
from .scheduler import Scheduler  # noqa: F401


def __getattr__(name):
    # model / pipeline classes need torch + the HIP library; import them lazily so that the pure
    # host pieces (scheduler, weight tables) stay importable anywhere
    if name in ("StableDiffusion", "StableDiffusionBase"):
        from . import stable_diffusion as m
        return getattr(m, name)
    if name in ("DiffusionModel", "ImageDecoder", "ImageEncoder", "ControlNet", "HintNet", "TextEncoder", "TextClipEmbedding"):
        from . import models as m
        return getattr(m, name)
    if name == "shutdown":   # destroy every captured hipGraph and drain the device (also runs at interpreter exit)
        from ._lib import shutdown
        return shutdown
    raise AttributeError(name)
