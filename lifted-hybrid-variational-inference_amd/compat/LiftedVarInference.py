"""Drop-in alias of the reference module of the same name: re-exports lhvi.vi."""
from lhvi.vi import LiftedVarInference as VarInference  # noqa: F401  (the reference names both classes VarInference)
