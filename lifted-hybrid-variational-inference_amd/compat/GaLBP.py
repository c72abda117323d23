"""Drop-in alias of the reference module of the same name: re-exports lhvi.gabp."""
from lhvi.gabp import GaLBP  # noqa: F401
from lhvi.lifting import *  # noqa: F401,F403
