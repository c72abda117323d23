"""Drop-in alias of the reference module of the same name: re-exports lhvi.lifting."""
from lhvi.graph import *  # noqa: F401,F403
from lhvi.lifting import SuperRV, SuperF, CompressedGraph  # noqa: F401
