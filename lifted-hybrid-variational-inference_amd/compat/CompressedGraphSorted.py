"""Drop-in alias of the reference module of the same name: re-exports lhvi.lifting."""
from lhvi.lifting import SuperRV, SuperF, CompressedGraph as CompressedGraphSorted  # noqa: F401
