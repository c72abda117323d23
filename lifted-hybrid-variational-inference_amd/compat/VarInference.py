"""Drop-in alias of the reference module of the same name: re-exports lhvi.vi."""
from lhvi.vi import VarInference  # noqa: F401
