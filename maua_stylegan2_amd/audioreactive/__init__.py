"""Drop-in for the reference's ``audioreactive`` package: everything star-exported as ``ar.*``
(/root/reference/audioreactive/__init__.py:1-5)."""
from . import signal as _signal_module
from .bend import *  # noqa: F401,F403
from .latent import *  # noqa: F401,F403
from .signal import *  # noqa: F401,F403
from .signal import set_SMF  # noqa: F401
from ..models.stylegan2 import Generator  # noqa: F401,E402  (the reference's latent.py leaks it into ar.* through its star import)
from .util import *  # noqa: F401,F403,E402  (diagnostics: info + matplotlib plots, imported lazily)
