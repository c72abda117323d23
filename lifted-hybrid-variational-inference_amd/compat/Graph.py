"""Drop-in alias of the reference module of the same name: re-exports lhvi.graph."""
from lhvi.graph import *  # noqa: F401,F403
from lhvi.graph import Domain, Potential, RV, F, Graph  # noqa: F401
from numpy import linspace  # noqa: F401  (the reference module exports it; Demo generators rely on that)
import numpy as np  # noqa: F401
