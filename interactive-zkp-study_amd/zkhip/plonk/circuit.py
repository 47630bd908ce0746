"""Gate-list circuits for PLONK (mirrors zkp/plonk/circuit.py): selectors q_L q_R q_O q_M q_C per gate,
copy constraints as (gate, wire) pairs, wire index 0/1/2 = a/b/c.  Host-side bookkeeping only."""
from ..field import FR, CURVE_ORDER

_MINUS_ONE = CURVE_ORDER - 1


class Gate:
    """q_L*a + q_R*b + q_O*c + q_M*a*b + q_C = 0  (circuit.py Gate)."""

    def __init__(self, q_l, q_r, q_o, q_m, q_c):
        self.q_l, self.q_r, self.q_o, self.q_m, self.q_c = (v if isinstance(v, FR) else FR(v) for v in (q_l, q_r, q_o, q_m, q_c))

    def check(self, a, b, c):
        a, b, c = FR(a), FR(b), FR(c)
        return self.q_l * a + self.q_r * b + self.q_o * c + self.q_m * (a * b) + self.q_c == FR(0)


class Circuit:
    def __init__(self):
        self.gates = []
        self.copy_constraints = []
        self.num_public_inputs = 0

    @property
    def n(self):
        return len(self.gates)

    def _push(self, gate):
        self.gates.append(gate)
        return len(self.gates) - 1

    def add_multiplication_gate(self):
        return self._push(Gate(0, 0, _MINUS_ONE, 1, 0))          # a*b = c

    def add_addition_gate(self):
        return self._push(Gate(1, 1, _MINUS_ONE, 0, 0))          # a+b = c

    def add_constant_gate(self, constant):
        return self._push(Gate(1, 0, _MINUS_ONE, 0, constant))   # a+const = c

    def add_public_input_gate(self):
        self.num_public_inputs += 1
        return self._push(Gate(0, 0, 1, 0, 0))

    def add_copy_constraint(self, gate1, wire1, gate2, wire2):
        self.copy_constraints.append((gate1, wire1, gate2, wire2))

    def get_selector_polynomials(self):
        """Five evaluation vectors, one entry per gate (circuit.py get_selector_polynomials)."""
        return tuple([getattr(g, name) for g in self.gates] for name in ("q_l", "q_r", "q_o", "q_m", "q_c"))

    def build_copy_constraints(self):
        """Permutation over the 3n wire positions (position = wire*n + gate): start from the identity
        and swap the images of the two positions of every copy constraint, which merges their cycles."""
        n = self.n
        sigma = list(range(3 * n))
        for g1, w1, g2, w2 in self.copy_constraints:
            p1, p2 = w1 * n + g1, w2 * n + g2
            sigma[p1], sigma[p2] = sigma[p2], sigma[p1]
        return sigma

    def compute_witness(self, assignments):
        """Left to subclasses / factory methods, as in the reference (circuit.py:238-252)."""
        raise NotImplementedError("implement in a subclass or use a factory method such as x3_plus_x_plus_5_eq_35()")

    @staticmethod
    def x3_plus_x_plus_5_eq_35():
        """The reference's toy circuit x^3 + x + 5 = 35 at x = 3 (circuit.py:286-331):
        -> (circuit, a_vals, b_vals, c_vals, public_inputs)."""
        c = Circuit()
        c.add_multiplication_gate()     # x * x   = x^2
        c.add_multiplication_gate()     # x^2 * x = x^3
        c.add_addition_gate()           # x^3 + x
        c.add_constant_gate(5)          # (x^3 + x) + 5
        for g1, w1, g2, w2 in ((0, 0, 0, 1), (0, 0, 1, 1), (0, 0, 2, 1), (0, 2, 1, 0), (1, 2, 2, 0), (2, 2, 3, 0)):
            c.add_copy_constraint(g1, w1, g2, w2)
        x = FR(3)
        x2, x3 = x * x, x * x * x
        s = x3 + x
        c.num_public_inputs = 1
        return c, [x, x2, x3, s], [x, x, x, FR(0)], [x2, x3, s, s + FR(5)], [FR(35)]
