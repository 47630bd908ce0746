"""CPU test of the oracle-side expectations the -m gpu Groth16-at-scale tests compare against (tests/helpers.py:
chain_closed_form_oracle, chain_crs_scalars): at small sizes they must equal the definition evaluated the slow way --
Lagrange basis polynomials as explicit products, the QAP sums wire by wire (zkp/groth16/test.py:303-325,
zkp/groth16/setup.py:42-60) -- and the host glue of the product (ChainCircuit.qap_at, closed_form_scalars)."""
import pytest

import py_ref as pr
from helpers import chain_closed_form_oracle, chain_crs_scalars, chain_witness, lagrange_at
from zkhip.groth16.prover_ntt import ChainCircuit

R = pr.R
TOXIC = dict(alpha=3926, beta=3604, gamma=2971, delta=1357, x=3721 + (1 << 200))


def _lagrange_products(m, x):
    w = pr.get_root_of_unity(m)
    pts = [pow(w, j, R) for j in range(m)]
    out = []
    for k in range(m):
        num = den = 1
        for j in range(m):
            if j != k:
                num = num * (x - pts[j]) % R
                den = den * (pts[k] - pts[j]) % R
        out.append(num * pow(den, -1, R) % R)
    return out


@pytest.mark.parametrize("log_m", [1, 3, 5])
def test_chain_expectations_equal_the_definition(log_m):
    circ = ChainCircuit(log_m, seed=11)
    m, W = circ.m, circ.num_wires
    w = chain_witness(circ.consts, circ.t0)
    assert w == circ.witness()[0]
    x, al, be, de = TOXIC["x"], TOXIC["alpha"], TOXIC["beta"], TOXIC["delta"]
    L = _lagrange_products(m, x)
    assert [lagrange_at(m, k, x) for k in range(m)] == L
    # per-wire QAP values from the R1CS rows: A = B = e_{1+k}; C = e_{2+k} - e_{1+k} - c_k e_0
    Ai, Ci = [0] * W, [0] * W
    for k in range(m):
        Ai[1 + k] = (Ai[1 + k] + L[k]) % R
        Ci[2 + k] = (Ci[2 + k] + L[k]) % R
        Ci[1 + k] = (Ci[1 + k] - L[k]) % R
        Ci[0] = (Ci[0] - circ.consts[k] * L[k]) % R
    Bi = Ai
    assert (Ai, Bi, Ci) == tuple(circ.qap_at(x)[:3])
    zx = (pow(x, m, R) - 1) % R
    dinv = pow(de, -1, R)
    r, s = 4106, 4565
    a_x = sum(w[i] * Ai[i] for i in range(W)) % R
    c_x = sum(w[i] * Ci[i] for i in range(W)) % R
    h_x = (a_x * a_x - c_x) * pow(zx, -1, R) % R
    A = (al + a_x + r * de) % R
    B = (be + a_x + s * de) % R
    priv = sum(w[i] * ((be * Ai[i] + al * Bi[i] + Ci[i]) % R) for i in range(2, W)) % R * dinv % R
    C = (priv + h_x * zx % R * dinv + s * A + r * B - r * s * de) % R
    assert chain_closed_form_oracle(circ.consts, w, TOXIC, r, s) == (A, B, C)
    i14 = list(range(2, W))
    s12, s14, s15 = chain_crs_scalars(circ.consts, TOXIC, [0, m - 1], i14, [0, max(m - 2, 0)])
    assert s12 == [1, pow(x, m - 1, R)]
    assert s14 == [(be * Ai[i] + al * Bi[i] + Ci[i]) % R * dinv % R for i in i14]
    assert s15 == [zx * dinv % R, pow(x, max(m - 2, 0), R) * zx % R * dinv % R]
