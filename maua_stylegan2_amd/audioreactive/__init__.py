"""Drop-in for the reference's ``audioreactive`` package: everything star-exported as ``ar.*``
(/root/reference/audioreactive/__init__.py:1-5)."""
from . import signal as _signal_module
from .bend import *  # noqa: F401,F403
from .latent import *  # noqa: F401,F403
from .signal import *  # noqa: F401,F403
from .signal import set_SMF  # noqa: F401
from ..models.stylegan2 import Generator  # noqa: F401,E402  (the reference's latent.py leaks it into ar.* through its star import)


def _plotting_out_of_scope(name):
    def stub(*args, **kwargs):
        import warnings

        warnings.warn(f"ar.{name}: the matplotlib debugging helpers of audioreactive/util.py are out of scope of this build "
                      "(SURVEY.md §2 row 10); the call is ignored so that unmodified plugins (examples/kelp.py:33) keep running")

    stub.__name__ = name
    return stub


for _name in ("info", "plot_signals", "plot_spectra", "plot_audio", "plot_chroma_comparison"):
    globals()[_name] = _plotting_out_of_scope(_name)
