"""Drop-in alias of the reference module of the same name: re-exports lhvi.utils."""
from lhvi.utils import *  # noqa: F401,F403
