import sys; sys.path.insert(0, 'lifted-hybrid-variational-inference_amd')
import numpy as np
from lhvi.graph import Domain, RV, F, Graph
from lhvi.potentials import GaussianPotential, X2Potential
from lhvi.pbp import EPBP
from lhvi.gabp import GaBP
d = Domain((-10, 10), continuous=True, integral_points=np.linspace(-10, 10, 32))
a, b, c = RV(d), RV(d), RV(d, value=1.5)
g = Graph()
g.rvs = [a, b, c]
g.factors = [F(GaussianPotential([0., 0.], [[2., 1.], [1., 2.]]), nb=[a, b]), F(GaussianPotential([0., 0.], [[2., -1.], [-1., 2.]]), nb=[b, c]),
             F(X2Potential(1., 4.), nb=[a])]
g.init_nb()
np.random.seed(0)
bp = EPBP(g, n=64, proposal_approximation='simple'); bp.run(10)
print(bp.map(a), bp.belief(0.3, b))
gbp = GaBP(g); gbp.run(20); print(gbp.get_belief_params(a))
mp, lb = bp.map_all(); print(mp)
