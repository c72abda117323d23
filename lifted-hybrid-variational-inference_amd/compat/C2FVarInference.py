"""``C2FVarInference`` under the reference's module name (``from C2FVarInference import VarInference``)."""
from lhvi.c2fvi import *  # noqa: F401,F403
from lhvi.c2fvi import VarInference  # noqa: F401
