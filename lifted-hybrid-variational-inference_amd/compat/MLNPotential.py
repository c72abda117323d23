"""Drop-in alias of the reference module of the same name: re-exports lhvi.mln."""
from lhvi.mln import *  # noqa: F401,F403
from math import e  # noqa: F401
import numpy as np  # noqa: F401
