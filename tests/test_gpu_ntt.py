"""GPU parity tests (-m gpu): the HIP NTT, through the C ABI, bit-exact against the oracle and the
golden fixtures; round trips and Horner spot checks at the BASELINE.json size (2^22)."""
import json
import os

import numpy as np
import pytest

import c_oracle as co
import py_ref as o
from helpers import limb_row, rand_fr_limbs
from zkhip import _lib
from zkhip.device import NttPlan, fr_quotient

pytestmark = pytest.mark.gpu


def ntt(X, inverse=False, k=None):
    d = np.array(X, dtype=np.uint64, copy=True)
    L = d.shape[0].bit_length() - 1
    kk = None if k is None else co.to_limbs([k])
    rc = _lib.load().zk_ntt_fr(_lib.ptr(d), L, 1 if inverse else 0, None if kk is None else _lib.ptr(kk))
    assert rc == 0, _lib.load().zk_last_error()
    return d


@pytest.mark.parametrize("L", list(range(0, 21)))
def test_ntt_bit_exact_vs_oracle(L):
    """Every pass structure: 1 pass (L <= 10), 2 passes (11..16), 3 passes (17..24)."""
    rng = np.random.default_rng(300 + L)
    n = 1 << L
    X = rand_fr_limbs(rng, n) if L <= 14 else _fast_rand(rng, n)
    w = o.get_root_of_unity(n)
    Y = ntt(X)
    assert np.array_equal(Y, co.ntt_arr(X, w))
    assert np.array_equal(ntt(Y, inverse=True), X)
    assert np.array_equal(ntt(X, inverse=True), co.ntt_arr(X, w, inverse=True))


def _fast_rand(rng, n):
    from zkhip.synthetic import random_scalars
    return random_scalars(rng, n)


def test_ntt_golden_fixtures(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "ntt_small.json")))
    for name, c in g["cases"].items():
        coeffs = [int(v) for v in c["coeffs"]]
        X = co.to_limbs([v % o.R for v in coeffs])
        assert co.from_limbs(ntt(X)) == [int(v) for v in c["fft"]], name
        assert co.from_limbs(ntt(X, inverse=True)) == [int(v) for v in c["ifft_of_coeffs"]], name
        assert co.from_limbs(ntt(X, k=5)) == [int(v) for v in c["coset_fft_k5"]], name
        assert co.from_limbs(ntt(ntt(X, k=5), inverse=True, k=5)) == [v % o.R for v in coeffs], name


@pytest.mark.parametrize("L,k", [(3, 5), (10, 5), (12, 7), (17, 5)])
def test_coset_ntt_vs_oracle(L, k):
    rng = np.random.default_rng(400 + L)
    n = 1 << L
    X = rand_fr_limbs(rng, n) if L <= 12 else _fast_rand(rng, n)
    w = o.get_root_of_unity(n)
    # coset_fft = scale by k^i then fft (zkp/plonk/utils.py:145-176); scale with Python ints
    xs = co.from_limbs(X)
    kp, scaled = 1, []
    for v in xs:
        scaled.append(v * kp % o.R)
        kp = kp * k % o.R
    exp = co.ntt_arr(co.to_limbs(scaled), w)
    got = ntt(X, k=k)
    assert np.array_equal(got, exp)
    assert np.array_equal(ntt(got, inverse=True, k=k), X)


@pytest.mark.parametrize("L,m", [(0, 0), (0, 1), (3, 5), (10, 1), (10, 700), (12, 1024), (12, 4096), (17, 40000), (18, 0)])
def test_padded_transform_equals_the_zero_filled_one(L, m):
    """zk_ntt_dev_padded: an input that is zero from element m on, read from one buffer and written to another (or in place), with
    and without the coset shift, both directions -- the same vectors as the in-place transform of the explicitly zero-padded copy
    (itself checked against the oracle above).  The elements behind m hold garbage that must not be read."""
    import torch
    rng = np.random.default_rng(500 + L + m)
    n = 1 << L
    X = _fast_rand(rng, n)
    Z = X.copy()
    Z[m:] = 0
    d_in = torch.from_numpy(X.view(np.int64)).cuda()            # garbage (non-zero) behind m
    plan = NttPlan(L)
    st = torch.cuda.current_stream().cuda_stream
    for inverse in (False, True):
        for k in (None, 5):
            want = ntt(Z, inverse=inverse, k=k)
            d_out = torch.full((n, 4), -1, dtype=torch.int64, device="cuda")
            plan.run_padded(d_in.data_ptr(), d_out.data_ptr(), m, inverse, k, st)
            torch.cuda.synchronize()
            assert np.array_equal(d_out.cpu().numpy().view(np.uint64), want), (inverse, k)
            assert np.array_equal(d_in.cpu().numpy().view(np.uint64), X)          # the input is left alone
            d_same = d_in.clone()
            plan.run_padded(d_same.data_ptr(), d_same.data_ptr(), m, inverse, k, st)     # in place
            torch.cuda.synchronize()
            assert np.array_equal(d_same.cpu().numpy().view(np.uint64), want), ("in place", inverse, k)
    plan.close()


def test_ntt_edge_values_and_errors():
    n = 16
    X = co.to_limbs([0, 1, o.R - 1, o.R - 2] * 4)
    assert np.array_equal(ntt(X), co.ntt_arr(X, o.get_root_of_unity(n)))
    Z = np.zeros((n, 4), dtype=np.uint64)
    assert not ntt(Z).any()
    bad = X.copy()
    bad[3] = limb_row(o.R)
    assert _lib.load().zk_ntt_fr(_lib.ptr(bad), 4, 0, None) == _lib.ZK_ERR_INVALID
    assert _lib.load().zk_ntt_fr(_lib.ptr(X), 29, 0, None) == _lib.ZK_ERR_INVALID


def test_ntt_2pow22_round_trip_and_horner():
    """BASELINE.json configs[2]: 2^22 coefficients, forward + inverse, device-resident."""
    import torch
    rng = np.random.default_rng(15)
    L = 22
    n = 1 << L
    X = _fast_rand(rng, n)
    d = torch.from_numpy(X.view(np.int64)).cuda()
    st = torch.cuda.current_stream().cuda_stream
    plan = NttPlan(L)
    plan.run(d.data_ptr(), False, None, st)
    torch.cuda.synchronize()
    Y = d.cpu().numpy().view(np.uint64)
    w = o.get_root_of_unity(n)
    for i in (0, 1, 2, 1234567, n // 2, n - 1):
        assert co.from_limbs(Y[i:i + 1])[0] == co.fr_horner_arr(X, pow(w, i, o.R)), i
    plan.run(d.data_ptr(), True, None, st)
    torch.cuda.synchronize()
    assert np.array_equal(d.cpu().numpy().view(np.uint64), X)
    # linearity: NTT(x + y) = NTT(x) + NTT(y) on a slice-sized problem of the same plan family
    # coset round trip at full size
    plan.run(d.data_ptr(), False, 5, st)
    plan.run(d.data_ptr(), True, 5, st)
    torch.cuda.synchronize()
    assert np.array_equal(d.cpu().numpy().view(np.uint64), X)


@pytest.mark.parametrize("L", [22, 21, 23])
def test_ntt_2pow22_bit_exact_vs_oracle(L):
    """BASELINE.json configs[2] in full: every one of the 2^22 outputs against the oracle's recursive radix-2 NTT -- and the
    two neighbouring sizes, whose digit splits (8 + 7 + 6 and 8 + 8 + 7: one odd digit, i.e. a single-stage last round in one
    pass) the sizes up to 2^20 and 2^22 = 8 + 8 + 6 do not reach."""
    import torch
    rng = np.random.default_rng(2200 + L)
    n = 1 << L
    X = _fast_rand(rng, n)
    d = torch.from_numpy(X.view(np.int64).copy()).cuda()
    NttPlan(L).run(d.data_ptr(), False, None, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(d.cpu().numpy().view(np.uint64), co.ntt_arr(X, o.get_root_of_unity(n), False))


def test_ntt_2pow24_bit_exact_vs_oracle():
    """north_star's target size in full: all 2^24 outputs of the forward AND of the inverse transform against the oracle's
    recursive radix-2 NTT (8 + 8 + 8 digit split; polynomial.py:316-378 at a size the reference itself cannot reach)."""
    import torch
    L = 24
    n = 1 << L
    X = _fast_rand(np.random.default_rng(2424), n)
    w = o.get_root_of_unity(n)
    st = torch.cuda.current_stream().cuda_stream
    plan = NttPlan(L)
    for inverse in (False, True):
        d = torch.from_numpy(X.view(np.int64).copy()).cuda()
        plan.run(d.data_ptr(), inverse, None, st)
        torch.cuda.synchronize()
        assert np.array_equal(d.cpu().numpy().view(np.uint64), co.ntt_arr(X, w, inverse)), inverse
        del d
    plan.close()


def _scale_by_powers(X, k):
    """x_i * k^i mod r on the host (Python integers, blocks of 2^16 with a running power): the coset scaling of
    zkp/plonk/utils.py:145-176, independent of both the library and the C oracle."""
    out = np.empty_like(X)
    blk = 1 << 16
    kb = pow(k, blk, o.R)
    base = 1
    pw = [1] * blk
    for i in range(1, blk):
        pw[i] = pw[i - 1] * k % o.R
    for lo in range(0, X.shape[0], blk):
        xs = co.from_limbs(X[lo:lo + blk])
        out[lo:lo + blk] = co.to_limbs([v * p % o.R * base % o.R for v, p in zip(xs, pw)])
        base = base * kb % o.R
    return out


@pytest.mark.parametrize("L,k", [(20, 5), (20, 0x1D0F4C0FFEE), (22, 5), (22, 7)])
def test_coset_ntt_large_vs_oracle(L, k):
    """coset_fft / coset_ifft (utils.py:145-205) at the sizes the provers run them (PLONK's 4n-point quotient domain at 2^20 gates
    is 2^22), every output against the oracle: forward = the oracle's NTT of the host-scaled input; inverse = the host-unscaled
    oracle inverse NTT.  (Up to 2^17 in test_coset_ntt_vs_oracle; beyond that only through whole proofs until round 5.)"""
    import torch
    n = 1 << L
    X = _fast_rand(np.random.default_rng(4000 + L + k % 97), n)
    w = o.get_root_of_unity(n)
    st = torch.cuda.current_stream().cuda_stream
    plan = NttPlan(L)
    d = torch.from_numpy(X.view(np.int64).copy()).cuda()
    plan.run(d.data_ptr(), False, k, st)
    torch.cuda.synchronize()
    assert np.array_equal(d.cpu().numpy().view(np.uint64), co.ntt_arr(_scale_by_powers(X, k), w))
    d = torch.from_numpy(X.view(np.int64).copy()).cuda()
    plan.run(d.data_ptr(), True, k, st)
    torch.cuda.synchronize()
    assert np.array_equal(d.cpu().numpy().view(np.uint64), _scale_by_powers(co.ntt_arr(X, w, True), pow(k, -1, o.R)))
    plan.close()


def test_ntt_maximum_domain_2pow28_round_trip():
    """The largest domain the reference admits (get_root_of_unity: n <= 2^28, zkp/plonk/field.py:169-172): forward + inverse
    restores all 2^28 elements, and one output is checked against the closed form of a two-term polynomial."""
    import torch
    L = 28
    n = 1 << L
    rng = np.random.default_rng(28)
    base = _fast_rand(rng, 1 << 20)
    d = torch.from_numpy(base.view(np.int64)).cuda().repeat(n >> 20, 1)       # 8.6 GB on the device, built there
    ref = d.clone()
    st = torch.cuda.current_stream().cuda_stream
    plan = NttPlan(L)
    plan.run(d.data_ptr(), False, None, st)
    plan.run(d.data_ptr(), True, None, st)
    torch.cuda.synchronize()
    assert torch.equal(d, ref)
    del ref
    # x = a + b X^j  ->  y[k] = a + b w^(j k)
    d.zero_()
    a, b, j = 12345678901234567890, 987654321987654321, (1 << 27) + 12345
    d[0:1] = torch.from_numpy(co.to_limbs([a]).view(np.int64)).cuda()
    d[j:j + 1] = torch.from_numpy(co.to_limbs([b]).view(np.int64)).cuda()
    plan.run(d.data_ptr(), False, None, st)
    w = o.get_root_of_unity(n)
    for k in (0, 1, 77, n // 2 + 5, n - 1):
        got = co.from_limbs(d[k:k + 1].cpu().numpy().view(np.uint64))[0]
        assert got == (a + b * pow(w, j * k, o.R)) % o.R, k
    with pytest.raises(_lib.ZkhipError):
        NttPlan(29)                                                          # 2^29 is beyond the field's 2-adicity


def test_ntt_linearity_2pow18():
    rng = np.random.default_rng(16)
    n = 1 << 18
    X, Y = _fast_rand(rng, n), _fast_rand(rng, n)
    m = 2048
    # sum on a prefix in Python, zero elsewhere keeps the check cheap but exercises all passes
    Xs, Ys = X.copy(), Y.copy()
    Xs[m:] = 0
    Ys[m:] = 0
    s = co.to_limbs([(a + b) % o.R for a, b in zip(co.from_limbs(Xs[:m]), co.from_limbs(Ys[:m]))])
    Zs = np.zeros_like(Xs)
    Zs[:m] = s
    a, b, c = co.from_limbs(ntt(Xs)[:512]), co.from_limbs(ntt(Ys)[:512]), co.from_limbs(ntt(Zs)[:512])
    assert all((x + y) % o.R == z for x, y, z in zip(a, b, c))


def test_fr_quotient_kernel():
    import torch
    rng = np.random.default_rng(17)
    n = 5000
    A, B, C = rand_fr_limbs(rng, n), rand_fr_limbs(rng, n), rand_fr_limbs(rng, n)
    zinv = int.from_bytes(rng.bytes(31), "little")
    dA, dB, dC = (torch.from_numpy(v.view(np.int64)).cuda() for v in (A, B, C))
    dO = torch.empty_like(dA)
    fr_quotient(dO.data_ptr(), dA.data_ptr(), dB.data_ptr(), dC.data_ptr(), zinv, n, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = co.from_limbs(dO.cpu().numpy().view(np.uint64))
    a, b, c = co.from_limbs(A), co.from_limbs(B), co.from_limbs(C)
    assert got == [((x * y - z) * zinv) % o.R for x, y, z in zip(a, b, c)]


def _oracle_coset(X, n, inverse, k):
    """coset_fft / coset_ifft of the oracle (utils.py:145-205): scale by k^i before a forward transform, by k^-i after an inverse."""
    w = o.get_root_of_unity(n)
    if k is None:
        return co.ntt_arr(X, w, inverse)
    if not inverse:
        vals = co.from_limbs(X)
        return co.ntt_arr(co.to_limbs([v * pow(k, i, o.R) % o.R for i, v in enumerate(vals)]), w, False)
    vals = co.from_limbs(co.ntt_arr(X, w, True))
    kinv = pow(k, -1, o.R)
    return co.to_limbs([v * pow(kinv, i, o.R) % o.R for i, v in enumerate(vals)])


@pytest.mark.parametrize("L,jobs,inverse,k,short", [(0, 3, False, None, False), (1, 2, True, None, False), (5, 3, False, 5, False), (8, 4, True, 5, True),
                                                    (9, 3, False, None, True), (12, 3, True, None, False), (13, 2, False, 7, True), (16, 4, False, None, False),
                                                    (17, 3, True, 5, False)])
def test_ntt_multi_vs_oracle(L, jobs, inverse, k, short):
    """zk_ntt_dev_multi: up to four transforms of separate buffers in one launch per pass -- one, two and three passes, both
    directions, with and without a coset shift, zero-padded inputs (in_len < n), job 0 in place and the others out of place --
    every output against the oracle's transform of that job alone, and the inputs of the out-of-place jobs untouched."""
    import torch
    rng = np.random.default_rng(7700 + 31 * L + jobs)
    n = 1 << L
    in_len = max(1, (n * 3) // 4) if short else n
    st = torch.cuda.current_stream().cuda_stream
    plan = NttPlan(L)
    Xs, want = [], []
    for b in range(jobs):
        X = rand_fr_limbs(rng, n) if L <= 12 else _fast_rand(rng, n)
        Xs.append(X)
        Z = X.copy()
        Z[in_len:] = 0
        want.append(_oracle_coset(Z, n, inverse, k))
    d_in = [torch.from_numpy(X.view(np.int64).copy()).cuda() for X in Xs]
    d_out = [d_in[0]] + [torch.full((n, 4), -1, dtype=torch.int64, device="cuda") for _ in range(jobs - 1)]
    plan.run_multi([(a.data_ptr(), b.data_ptr()) for a, b in zip(d_in, d_out)], in_len, inverse, k, st)
    torch.cuda.synchronize()
    for b in range(jobs):
        assert np.array_equal(d_out[b].cpu().numpy().view(np.uint64), want[b]), (L, b)
        if b:
            assert np.array_equal(d_in[b].cpu().numpy().view(np.uint64), Xs[b]), (L, b)
    plan.close()


def test_ntt_multi_at_2pow20_and_refusals():
    """Three 2^20-point transforms in one launch per pass (what the Groth16 prover's groups are) against the oracle in full; a buffer
    written by one job and used by another, more than four jobs and a null buffer are refused."""
    import ctypes
    import torch
    L, n = 20, 1 << 20
    st = torch.cuda.current_stream().cuda_stream
    plan = NttPlan(L)
    Xs = [_fast_rand(np.random.default_rng(7800 + b), n) for b in range(3)]
    w = o.get_root_of_unity(n)
    d_in = [torch.from_numpy(X.view(np.int64).copy()).cuda() for X in Xs]
    d_out = [torch.empty((n, 4), dtype=torch.int64, device="cuda") for _ in range(3)]
    for inverse in (False, True):
        plan.run_multi([(a.data_ptr(), b.data_ptr()) for a, b in zip(d_in, d_out)], n, inverse, None, st)
        torch.cuda.synchronize()
        for b in range(3):
            assert np.array_equal(d_out[b].cpu().numpy().view(np.uint64), co.ntt_arr(Xs[b], w, inverse)), (inverse, b)
    lib = _lib.load()

    def call(pairs):
        ins = (ctypes.c_void_p * len(pairs))(*[p[0] for p in pairs])
        outs = (ctypes.c_void_p * len(pairs))(*[p[1] for p in pairs])
        return lib.zk_ntt_dev_multi(plan._h, len(pairs), ins, outs, n, 0, None, st)
    a, b, c = (t.data_ptr() for t in d_out)
    assert call([(a, b), (b, c)]) != 0             # b written by job 0, read by job 1
    assert call([(a, c), (b, c)]) != 0             # c written twice
    assert call([(a, a)] * 5) != 0                 # more than four
    assert call([(a, a), (b, None)]) != 0          # null output
    assert call([(a, a + 64), (b, b)]) != 0        # input and output of one job overlap without being the same buffer
    assert call([(a, a), (b, b)]) == 0             # in place is fine
    torch.cuda.synchronize()
    plan.close()
