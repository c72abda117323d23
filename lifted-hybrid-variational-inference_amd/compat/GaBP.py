"""Drop-in alias of the reference module of the same name: re-exports lhvi.gabp."""
from lhvi.gabp import GaBP  # noqa: F401
