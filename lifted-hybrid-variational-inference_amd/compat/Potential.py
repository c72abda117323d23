"""Drop-in alias of the reference module of the same name: re-exports lhvi.potentials."""
from lhvi.potentials import *  # noqa: F401,F403
