"""Drop-in alias of the reference module of the same name: re-exports lhvi.pbp."""
from lhvi.lifting import SuperRV, SuperF, CompressedGraph  # noqa: F401
from lhvi.pbp import HybridLBP  # noqa: F401
