"""Copy-constraint permutation polynomials and the grand-product accumulator
(mirrors zkp/plonk/permutation.py:40-137); coset labels K1 = 2, K2 = 3."""
from ..field import FR, CURVE_ORDER

K1 = FR(2)
K2 = FR(3)


def _label(pos, n, domain):
    """Field label of wire position pos: omega^i, K1*omega^i, K2*omega^i for the a, b, c columns."""
    if pos < n:
        return domain[pos]
    if pos < 2 * n:
        return K1 * domain[pos - n]
    return K2 * domain[pos - 2 * n]


def build_permutation_polynomials(sigma, n, domain):
    """Evaluations of S_sigma1..3 on the domain (permutation.py:43-86)."""
    return tuple([_label(sigma[col * n + i], n, domain) for i in range(n)] for col in range(3))


def compute_accumulator(a_vals, b_vals, c_vals, sigma, n, domain, beta, gamma):
    """z(omega^0) = 1, z(omega^(i+1)) = z(omega^i) * num_i / den_i (permutation.py:89-137).
    All n-1 denominators are inverted together (one field inversion instead of n-1)."""
    s1, s2, s3 = build_permutation_polynomials(sigma, n, domain)
    r = CURVE_ORDER
    be, ga = int(beta), int(gamma)
    nums, dens = [], []
    for i in range(n - 1):
        a, b, c, w = int(a_vals[i]), int(b_vals[i]), int(c_vals[i]), int(domain[i])
        nums.append((a + be * w + ga) * (b + be * 2 * w + ga) % r * (c + be * 3 * w + ga) % r)
        dens.append((a + be * int(s1[i]) + ga) * (b + be * int(s2[i]) + ga) % r * (c + be * int(s3[i]) + ga) % r)
    # batch inversion with py_ecc's convention inv0(0) = 0 (FR(x) / FR(0) is FR(0) in the reference): a zero denominator is
    # left out of the running product and its "inverse" is 0, so z is 0 from that row on, as in the reference's loop
    pref = [1]
    for d in dens:
        pref.append(pref[-1] * (d or 1) % r)
    inv = pow(pref[-1], -1, r)
    inv_dens = [0] * len(dens)
    for i in range(len(dens) - 1, -1, -1):
        inv_dens[i] = pref[i] * inv % r if dens[i] else 0
        inv = inv * (dens[i] or 1) % r
    z = [1]
    for i in range(n - 1):
        z.append(z[-1] * nums[i] % r * inv_dens[i] % r)
    return [FR(v) for v in z]
