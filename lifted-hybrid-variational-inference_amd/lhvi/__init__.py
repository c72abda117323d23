"""lhvi -- MI355X-native message passing behind the API of leodd/Lifted-Hybrid-Variational-Inference.

Host object model (``graph``, ``potentials``, ``mln``, ``relational``, ``lifting``) mirrors the
reference's flat modules; solvers (``gabp``, ``pbp``, ``vi``) flatten it to CSR arrays and run the
sweeps as hand-written HIP kernels through the C ABI in ``include/lhvi.h``.
"""
from .graph import Domain, F, Graph, Potential, RV  # noqa: F401

__version__ = '0.1.0'
