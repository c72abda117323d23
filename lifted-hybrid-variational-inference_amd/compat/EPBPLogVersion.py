"""Drop-in alias of the reference module of the same name: re-exports lhvi.pbp."""
from lhvi.graph import *  # noqa: F401,F403
from lhvi.pbp import EPBP  # noqa: F401
