"""allophant_amd: MI355X-native acoustic-encoder forward path for kgnlp/allophant's ``Estimator.predict``.

Hand-written HIP kernels for gfx950 behind a C ABI (``include/allophant_amx.h``), wrapped by a Python façade that keeps
the reference's ``Estimator`` / ``Batch`` / ``Predictions`` names and shapes.  There is no CPU fallback: using the
compute path without the built ``liballophant_amx.so`` raises.
"""
from . import spec, synthetic  # noqa: F401

__all__ = ["spec", "synthetic", "Estimator", "Batch", "Predictions", "GreedyCTCDecoder", "CTCHypothesis"]
__version__ = "0.1.0"


def __getattr__(name):
    # the façade classes live in .estimator; resolved on first use so that importing the package stays light
    if name in ("Estimator", "Batch", "Predictions", "GreedyCTCDecoder", "CTCHypothesis"):
        from . import estimator

        return getattr(estimator, name)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
