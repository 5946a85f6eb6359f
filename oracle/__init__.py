"""CPU oracle (test infrastructure only; see oracle/spectral.py header)."""
from .spectral import *  # noqa: F401,F403
