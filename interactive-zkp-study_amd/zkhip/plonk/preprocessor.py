"""Circuit preprocessing (mirrors zkp/plonk/preprocessor.py:59-130): pad to a power of two, interpolate
the five selector and three permutation polynomials (8 inverse NTTs on the GPU) and commit to them
(8 G1 MSMs on the GPU)."""
from ..field import FR, get_root_of_unity, get_roots_of_unity
from .circuit import Gate
from .kzg import commit
from .permutation import build_permutation_polynomials
from .polynomial import Polynomial
from .utils import next_power_of_2


class PreprocessedData:
    pass


def preprocess(circuit, srs):
    pp = PreprocessedData()
    n = next_power_of_2(circuit.n)
    while len(circuit.gates) < n:                       # the reference pads the caller's circuit in place
        circuit.gates.append(Gate(0, 0, 0, 0, 0))
    pp.n = n
    pp.omega = get_root_of_unity(n)
    pp.domain = get_roots_of_unity(n)
    names = ("q_l", "q_r", "q_o", "q_m", "q_c")
    for name, evals in zip(names, circuit.get_selector_polynomials()):
        poly = Polynomial.from_evaluations(evals, pp.omega)
        setattr(pp, name + "_poly", poly)
        setattr(pp, name + "_comm", commit(poly, srs))
    pp.sigma = circuit.build_copy_constraints()
    for k, evals in enumerate(build_permutation_polynomials(pp.sigma, n, pp.domain), start=1):
        poly = Polynomial.from_evaluations(evals, pp.omega)
        setattr(pp, "s_sigma%d_poly" % k, poly)
        setattr(pp, "s_sigma%d_comm" % k, commit(poly, srs))
    pp.num_public_inputs = circuit.num_public_inputs
    return pp
