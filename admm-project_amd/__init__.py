"""admm-project_amd: MI355X-native ADMM iteration engine behind the admm()/getproxops() surface.

Host side (Python) mirrors the reference's MATLAB interface for the hot path only; all
per-iteration arithmetic runs in hand-written HIP kernels (csrc/) behind a plain C ABI
(include/admm_engine.h -> libadmm_hip.so).  No CPU fallback exists.
"""
from . import _lib, errorcheck, synth  # noqa: F401
from ._lib import AdmmError  # noqa: F401
from .api import ProxOp, admm, getproxops  # noqa: F401
from .engine import Engine  # noqa: F401
from . import testers  # noqa: F401,E402
from .solvers import (basispursuit, huberfit, lad, lasso, linearprogram, linearsvm, model, quadraticprogram,  # noqa: F401
                      totalvariation, totalvariation2d, unwrappedadmm)

__version__ = "0.1.0"
