"""Drop-in alias of the reference module of the same name: re-exports lhvi.relational."""
from lhvi.graph import *  # noqa: F401,F403
from lhvi.relational import *  # noqa: F401,F403
from numpy import linspace  # noqa: F401
import numpy as np  # noqa: F401
