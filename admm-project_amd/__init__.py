"""admm-project_amd: MI355X-native ADMM iteration engine behind the admm()/getproxops() surface."""
from . import synth  # noqa: F401
