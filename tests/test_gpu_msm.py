"""GPU parity tests (run with -m gpu on an MI355X): the HIP MSM pipeline, called through the C ABI,
against the oracle on the same seeded inputs -- bit-exact -- plus size-independent properties at the
BASELINE.json size (2^20)."""
import ctypes

import numpy as np
import pytest

import c_oracle as co
import py_ref as o
from helpers import arithmetic_g1_points, limb_row, rand_fr_limbs, rand_g1_limbs, rand_g2_limbs
from zkhip import _lib
from zkhip.device import MsmPlan
from zkhip.distributed import fold_partials

pytestmark = pytest.mark.gpu


def msm_g1(S, Pts):
    out = np.zeros(8, dtype=np.uint64)
    inf = ctypes.c_int(-1)
    rc = _lib.load().zk_msm_g1(_lib.ptr(S), _lib.ptr(Pts), S.shape[0], _lib.ptr(out), ctypes.byref(inf))
    assert rc == 0, _lib.load().zk_last_error()
    assert inf.value == (0 if out.any() else 1)
    return out


def oracle_g1(S, Pts):
    """Oracle MSM: per-term double-and-add for small inputs, the serial bucket method (checked against the former in
    tests/test_oracle.py) where that would take tens of seconds."""
    return co.g1_msm_arr(S, Pts) if S.shape[0] < 4096 else co.g1_msm_bucket_arr(S, Pts, 12)


def msm_g2(S, Pts):
    out = np.zeros(16, dtype=np.uint64)
    inf = ctypes.c_int(-1)
    rc = _lib.load().zk_msm_g2(_lib.ptr(S), _lib.ptr(Pts), S.shape[0], _lib.ptr(out), ctypes.byref(inf))
    assert rc == 0, _lib.load().zk_last_error()
    return out


# sizes straddle the window-width switches (c = 8 | 15 | 16 at n = 2^9 / 2^17) and block edges
@pytest.mark.parametrize("n", [1, 2, 3, 17, 255, 256, 257, 512, 513, 1000, 2048, 2049, 4095, 4096, 4097, 16384, 16385, 40000])
def test_g1_msm_bit_exact_vs_oracle(n):
    rng = np.random.default_rng(1000 + n)
    S = rand_fr_limbs(rng, n)
    Pts, _ = rand_g1_limbs(rng, n)
    assert np.array_equal(msm_g1(S, Pts), oracle_g1(S, Pts))


def test_g1_msm_edge_scalars_and_points():
    rng = np.random.default_rng(7)
    n = 600
    S = rand_fr_limbs(rng, n)
    Pts, _ = rand_g1_limbs(rng, n)
    S[0] = 0
    S[1] = limb_row(1)
    S[2] = limb_row(o.R - 1)
    S[3] = limb_row(2)
    S[4] = limb_row((1 << 253) + 12345)
    Pts[5] = 0                       # infinity input (Python None)
    Pts[7] = Pts[6]                  # duplicate point, different scalars
    Pts[9] = Pts[8]
    S[9] = S[8]                      # duplicate (scalar, point) pair -> doubling inside a bucket
    neg = co.g1_to_arr([o.g1_neg(co.g1_from_arr(Pts[10])[0])])[0]
    Pts[11] = neg
    S[11] = S[10]                    # P and -P with equal scalars cancel
    assert np.array_equal(msm_g1(S, Pts), co.g1_msm_arr(S, Pts))


def test_g1_msm_infinity_results():
    rng = np.random.default_rng(8)
    Pts, _ = rand_g1_limbs(rng, 10)
    Z = np.zeros((10, 4), dtype=np.uint64)
    assert not msm_g1(Z, Pts).any()                                  # all-zero scalars -> None
    S = rand_fr_limbs(rng, 10)
    assert not msm_g1(S, np.zeros((10, 8), dtype=np.uint64)).any()   # all-infinity points -> None
    # s*P + (r-s)*P = infinity
    s = int.from_bytes(rng.bytes(31), "little")
    S2 = co.to_limbs([s, o.R - s])
    P2 = np.stack([Pts[0], Pts[0]])
    assert not msm_g1(S2, P2).any()
    out = np.ones(8, dtype=np.uint64)
    inf = ctypes.c_int(0)
    assert _lib.load().zk_msm_g1(None, None, 0, _lib.ptr(out), ctypes.byref(inf)) == 0   # n = 0
    assert inf.value == 1 and not out.any()


def test_g1_msm_rejects_non_canonical_scalar():
    rng = np.random.default_rng(9)
    Pts, _ = rand_g1_limbs(rng, 4)
    S = rand_fr_limbs(rng, 4)
    S[2] = limb_row(o.R)  # == r: not canonical
    out = np.zeros(8, dtype=np.uint64)
    assert _lib.load().zk_msm_g1(_lib.ptr(S), _lib.ptr(Pts), 4, _lib.ptr(out), None) == _lib.ZK_ERR_INVALID


@pytest.mark.parametrize("pattern", ["all_equal", "witness_like", "tiny_values"])
def test_g1_msm_skewed_scalars(pattern):
    """Witness-like inputs put many points into very few buckets (SURVEY.md section 7 'hard parts')."""
    rng = np.random.default_rng(11)
    n = 6000
    Pts, _ = rand_g1_limbs(rng, n)
    if pattern == "all_equal":
        S = np.tile(rand_fr_limbs(rng, 1), (n, 1))
    elif pattern == "witness_like":
        S = rand_fr_limbs(rng, n)
        pick = rng.integers(0, 4, size=n)
        S[pick == 0] = 0
        S[pick == 1] = limb_row(1)
    else:
        S = co.to_limbs([int(v) for v in rng.integers(0, 40, size=n)])
    assert np.array_equal(msm_g1(S, Pts), co.g1_msm_arr(S, Pts))


@pytest.mark.parametrize("n", [1, 2, 33, 513, 700])
def test_g2_msm_bit_exact_vs_oracle(n):
    rng = np.random.default_rng(2000 + n)
    S = rand_fr_limbs(rng, n)
    Pts, _ = rand_g2_limbs(rng, n)
    if n > 4:
        S[0] = 0
        S[1] = limb_row(o.R - 1)
        Pts[2] = 0
        Pts[4] = Pts[3]
    assert np.array_equal(msm_g2(S, Pts), co.g2_msm_arr(S, Pts))


@pytest.mark.parametrize("pattern", ["all_equal", "witness_like"])
def test_g2_msm_skewed_scalars(pattern):
    """The heavy-bucket kernels and the multi-workgroup cell sort in their F_p^2 instantiation."""
    rng = np.random.default_rng(17)
    n = 3000
    base, _ = rand_g2_limbs(rng, 24)
    Pts = base[rng.integers(0, 24, size=n)]                      # few distinct points, many repeats: doublings inside buckets too
    if pattern == "all_equal":
        S = np.tile(rand_fr_limbs(rng, 1), (n, 1))
    else:
        S = rand_fr_limbs(rng, n)
        pick = rng.integers(0, 4, size=n)
        S[pick == 0] = 0
        S[pick == 1] = limb_row(1)
    assert np.array_equal(msm_g2(S, Pts), co.g2_msm_arr(S, Pts))


def _g2_bases(K):
    """P_i = k_i * G2 from the fixed-base kernel, a few of them checked against the oracle's k * G2."""
    n = K.shape[0]
    base = co.g2_to_arr([o.G2])
    Pts = np.zeros((n, 16), dtype=np.uint64)
    _lib.check(_lib.load().zk_fixed_base_g2(_lib.ptr(base), _lib.ptr(K), n, _lib.ptr(Pts)))
    idx = [0, 1, n // 2, n - 1]
    assert np.array_equal(Pts[idx], co.g2_fixed_base_arr(o.G2, K[idx]))
    return Pts


@pytest.mark.parametrize("n", [131072, 131073, (1 << 18) + 77, 1 << 20])
def test_g2_msm_unbound_closed_form_across_c16_switch(n):
    """proof_b's query (zkp/groth16/proving.py:35-45) at real sizes, UNBOUND bases: 15-bit windows up to 2^17 points, 16-bit
    windows above (msm_prepare_kernel<Fp2,16>, G2 cells of full size, the G2 window reduction).  P_i = k_i*G2, so the MSM is
    (sum s_i k_i) * G2 with the scalar and the point from the oracle (zkp/groth16/test.py:303-325 is this identity)."""
    from zkhip.synthetic import random_scalars
    rng = np.random.default_rng(7000 + n % 1000)
    S, K = random_scalars(rng, n), random_scalars(rng, n)
    S[0] = 0
    S[1] = limb_row(o.R - 1)
    Pts = _g2_bases(K)
    want = co.g2_mul(o.G2, co.fr_dot_arr(S, K))
    assert co.g2_from_arr(msm_g2(S, Pts))[0] == want
    if n == (1 << 18) + 77:
        # in full against the oracle's serial bucket method as well (a different restatement: unsigned windows, Jacobian)
        assert np.array_equal(msm_g2(S, Pts), co.g2_msm_bucket_arr(S, Pts, 14))


@pytest.mark.parametrize("pattern", ["witness_like", "all_equal", "forty_values"])
def test_g2_msm_skewed_scalars_large(pattern):
    """The G2 twin of test_g1_msm_skewed_scalars_large: hot digits at 2^18 points with 16-bit windows -- what a Groth16
    B query sees from a boolean-heavy witness (msm_heavy_kernel<Fp2>, the multi-workgroup cell sort) -- bit-exact against
    the oracle's serial bucket MSM, blocking and with three submissions in flight."""
    import torch
    from zkhip.synthetic import random_scalars
    rng = np.random.default_rng(4343)
    n = (1 << 18) + 77
    K = random_scalars(rng, n)
    Pts = _g2_bases(K)
    S = random_scalars(rng, n)
    pick = rng.random(n)
    if pattern == "witness_like":
        S[pick < 0.25] = 0
        S[(pick >= 0.25) & (pick < 0.5)] = limb_row(1)
    elif pattern == "all_equal":
        S[:] = S[0]
    else:
        S = S[rng.integers(0, 40, size=n)]
    want = co.g2_msm_bucket_arr(S, Pts, 14)
    assert co.g2_from_arr(want)[0] == co.g2_mul(o.G2, co.fr_dot_arr(S, K))       # the oracle against its own closed form
    assert np.array_equal(msm_g2(S, Pts), want)
    dS, dP = torch.from_numpy(S.view(np.int64)).cuda(), torch.from_numpy(Pts.view(np.int64)).cuda()
    st = torch.cuda.current_stream().cuda_stream
    plan = MsmPlan(_lib.GROUP_G2, n)
    tickets = [plan.submit(dS.data_ptr(), dP.data_ptr(), n, st) for _ in range(plan.max_in_flight())]
    for t in tickets:
        assert np.array_equal(plan.collect_limbs(t)[0], want)
    # bound bases (what the at-scale prover runs) must give the same point
    plan.bind(dP.data_ptr(), n, st)
    assert np.array_equal(plan.run_limbs(dS.data_ptr(), None, n, st)[0], want)
    plan.close()


def test_device_plan_reuse_partials_and_profile():
    import torch
    rng = np.random.default_rng(12)
    n = 20000
    S = rand_fr_limbs(rng, n)
    Pts, _ = rand_g1_limbs(rng, n)
    dS = torch.from_numpy(S.view(np.int64)).cuda()
    dP = torch.from_numpy(Pts.view(np.int64)).cuda()
    st = torch.cuda.current_stream().cuda_stream
    plan = MsmPlan(_lib.GROUP_G1, n)
    plan.set_profiling(True)
    full, inf = plan.run_limbs(dS.data_ptr(), dP.data_ptr(), n, st)
    assert not inf and np.array_equal(full, co.g1_msm_arr(S, Pts))
    ms = plan.stage_ms()
    assert len(ms) == 4 and all(v > 0 for v in ms)
    # the same plan serves smaller MSMs (other window widths) and sub-ranges
    for m in (300, 5000):
        got, _ = plan.run_limbs(dS.data_ptr(), dP.data_ptr(), m, st)
        assert np.array_equal(got, co.g1_msm_arr(S[:m], Pts[:m]))
    # chunk-additivity = the multi-GPU recombination: fold(partial(lo half), partial(hi half)) == full
    h = n // 2
    p0 = plan.run_partial(dS.data_ptr(), dP.data_ptr(), h, st)
    p1 = plan.run_partial(dS.data_ptr() + h * 32, dP.data_ptr() + h * 64, n - h, st)
    got = fold_partials(_lib.GROUP_G1, np.stack([p0, p1]))
    assert co.g1_to_arr([(int(got[0]), int(got[1]))])[0].tolist() == full.tolist()


def test_pipelined_submit_collect():
    """zk_msm_submit / zk_msm_collect: max_in_flight submissions outstanding (each in its own lane and stream),
    collected out of order, inputs overwritten in stream order right after submit, results independent of the overlap."""
    import torch
    rng = np.random.default_rng(21)
    n = 30000
    S1, S2, S3 = rand_fr_limbs(rng, n), rand_fr_limbs(rng, n), rand_fr_limbs(rng, n)
    Pts, _ = rand_g1_limbs(rng, n)
    d1, d2, d3 = (torch.from_numpy(S.view(np.int64)).cuda() for S in (S1, S2, S3))
    dP = torch.from_numpy(Pts.view(np.int64)).cuda()
    st = torch.cuda.current_stream().cuda_stream
    plan = MsmPlan(_lib.GROUP_G1, n)
    assert plan.max_in_flight() == 3
    t1 = plan.submit(d1.data_ptr(), dP.data_ptr(), n, st)
    d1.zero_()                                                     # stream-ordered reuse of an input buffer
    t2 = plan.submit(d2.data_ptr(), dP.data_ptr(), 5000, st)
    t3 = plan.submit(d3.data_ptr(), dP.data_ptr(), n, st)
    with pytest.raises(_lib.ZkhipError):
        plan.submit(d2.data_ptr(), dP.data_ptr(), n, st)          # a fourth one must wait for a collect
    r3, inf3 = plan.collect_limbs(t3)                              # out of order
    r1, inf1 = plan.collect_limbs(t1)
    t4 = plan.submit(d2.data_ptr(), dP.data_ptr(), 0, st)          # empty MSM through the same path
    r2, inf2 = plan.collect_limbs(t2)
    r4, inf4 = plan.collect_limbs(t4)
    assert np.array_equal(r1, oracle_g1(S1, Pts)) and not inf1
    assert np.array_equal(r2, co.g1_msm_arr(S2[:5000], Pts[:5000])) and not inf2
    assert np.array_equal(r3, oracle_g1(S3, Pts)) and not inf3
    assert inf4 and not r4.any()
    with pytest.raises(_lib.ZkhipError):
        plan.collect_limbs(t1)                                     # already collected
    # a long alternation keeps every lane busy; all results must equal the blocking call's
    want = [oracle_g1(S, Pts) for S in (S2, S3)]
    pend = []
    for i in range(12):
        pend.append((i % 2, plan.submit((d2, d3)[i % 2].data_ptr(), dP.data_ptr(), n, st)))
        if len(pend) == plan.max_in_flight():
            k, t = pend.pop(0)
            assert np.array_equal(plan.collect_limbs(t)[0], want[k])
    for k, t in pend:
        assert np.array_equal(plan.collect_limbs(t)[0], want[k])


def test_g1_msm_2pow16_bit_exact():
    rng = np.random.default_rng(13)
    n = 1 << 16
    S = rand_fr_limbs(rng, n)
    Pts, _ = rand_g1_limbs(rng, n)
    assert np.array_equal(msm_g1(S, Pts), oracle_g1(S, Pts))


@pytest.mark.parametrize("pattern", ["witness_like", "all_equal", "two_values", "minus_one_heavy", "forty_values"])
def test_g1_msm_skewed_scalars_large(pattern):
    """Hot digits at 2^18 points: cells far larger than one workgroup's share and buckets with up to n entries
    (multi-workgroup cell sort, heavy-bucket wavefront tasks) -- bit-exact against the oracle's serial bucket MSM."""
    from zkhip.synthetic import random_scalars
    rng = np.random.default_rng(4242)
    n = (1 << 18) + 77
    K = random_scalars(rng, n)
    g1 = np.array([[1, 0, 0, 0, 2, 0, 0, 0]], dtype=np.uint64)
    Pts = np.zeros((n, 8), dtype=np.uint64)
    _lib.check(_lib.load().zk_fixed_base_g1(_lib.ptr(g1), _lib.ptr(K), n, _lib.ptr(Pts)))
    S = random_scalars(rng, n)
    pick = rng.random(n)
    if pattern == "witness_like":
        S[pick < 0.25] = 0
        S[(pick >= 0.25) & (pick < 0.5)] = limb_row(1)
    elif pattern == "all_equal":
        S[:] = S[0]
    elif pattern == "two_values":
        S[pick < 0.5] = S[0]
        S[pick >= 0.5] = S[1]
    elif pattern == "forty_values":       # ~40 heavy buckets in every window, several wavefront tasks each: many per-bucket combines at once
        S = S[rng.integers(0, 40, size=n)]
    else:
        S[pick < 0.7] = limb_row(o.R - 1)
    want = co.g1_msm_bucket_arr(S, Pts, 14)
    assert np.array_equal(msm_g1(S, Pts), want)
    if pattern in ("forty_values", "witness_like"):
        # the same with three submissions in flight (the heavy-bucket kernel of one lane beside the others' accumulate kernels)
        import torch
        dS, dP = torch.from_numpy(S.view(np.int64)).cuda(), torch.from_numpy(Pts.view(np.int64)).cuda()
        st = torch.cuda.current_stream().cuda_stream
        plan = MsmPlan(_lib.GROUP_G1, n)
        tickets = [plan.submit(dS.data_ptr(), dP.data_ptr(), n, st) for _ in range(3)]
        for t in tickets:
            assert np.array_equal(plan.collect_limbs(t)[0], want)
        plan.close()


@pytest.mark.parametrize("n", [131072, 131073, 300001])
def test_g1_msm_closed_form_across_c16_switch(n):
    """P_i = k_i*G1 (generated on the GPU, spot-checked against the oracle): MSM = (sum s_i k_i) * G1."""
    from zkhip.synthetic import random_scalars
    rng = np.random.default_rng(n)
    S, K = random_scalars(rng, n), random_scalars(rng, n)
    g1 = np.array([[1, 0, 0, 0, 2, 0, 0, 0]], dtype=np.uint64)
    Pts = np.zeros((n, 8), dtype=np.uint64)
    _lib.check(_lib.load().zk_fixed_base_g1(_lib.ptr(g1), _lib.ptr(K), n, _lib.ptr(Pts)))
    idx = [0, n // 2, n - 1]
    assert np.array_equal(Pts[idx], co.g1_fixed_base_arr(o.G1, K[idx]))
    assert co.g1_from_arr(msm_g1(S, Pts))[0] == co.g1_mul(o.G1, co.fr_dot_arr(S, K))


def test_g1_msm_2pow20_closed_form_and_linearity():
    """BASELINE.json configs[1] size.  P_i = k_i*G1, so MSM(s, P) = (sum s_i k_i mod r) * G1; also
    MSM(2s) = 2*MSM(s) and MSM(s) + MSM(t) = MSM(s + t)."""
    from zkhip.synthetic import random_scalars
    rng = np.random.default_rng(14)
    n = 1 << 20
    S, T, K = random_scalars(rng, n), random_scalars(rng, n), random_scalars(rng, n)
    g1 = np.array([[1, 0, 0, 0, 2, 0, 0, 0]], dtype=np.uint64)
    Pts = np.zeros((n, 8), dtype=np.uint64)
    _lib.check(_lib.load().zk_fixed_base_g1(_lib.ptr(g1), _lib.ptr(K), n, _lib.ptr(Pts)))
    # spot-check the generated bases against the oracle
    idx = [0, 1, 12345, n - 1]
    assert np.array_equal(Pts[idx], co.g1_fixed_base_arr(o.G1, K[idx]))
    ms = msm_g1(S, Pts)
    assert co.g1_from_arr(ms)[0] == co.g1_mul(o.G1, co.fr_dot_arr(S, K))
    mt = msm_g1(T, Pts)
    assert co.g1_from_arr(mt)[0] == co.g1_mul(o.G1, co.fr_dot_arr(T, K))
    st_ints = [(a + b) % o.R for a, b in zip(co.from_limbs(S[:4096]), co.from_limbs(T[:4096]))]
    # linearity on a 4096-point prefix (full-size scalar addition in Python would dominate the test time)
    a = msm_g1(np.ascontiguousarray(S[:4096]), np.ascontiguousarray(Pts[:4096]))
    b = msm_g1(np.ascontiguousarray(T[:4096]), np.ascontiguousarray(Pts[:4096]))
    c = msm_g1(co.to_limbs(st_ints), np.ascontiguousarray(Pts[:4096]))
    assert co.g1_from_arr(c)[0] == co.g1_add(co.g1_from_arr(a)[0], co.g1_from_arr(b)[0])
    dbl = co.to_limbs([2 * v % o.R for v in co.from_limbs(S[:4096])])
    d = msm_g1(dbl, np.ascontiguousarray(Pts[:4096]))
    assert co.g1_from_arr(d)[0] == co.g1_add(co.g1_from_arr(a)[0], co.g1_from_arr(a)[0])


def test_g1_msm_2pow20_bit_exact_vs_oracle_pippenger():
    """BASELINE.json configs[1] in full: 2^20 random scalars x points, the GPU result against the oracle's serial
    bucket-method MSM (a structurally different restatement: unsigned windows, Jacobian, running sums) -- bit-exact."""
    from zkhip.synthetic import random_scalars
    rng = np.random.default_rng(2020)
    n = 1 << 20
    S, K = random_scalars(rng, n), random_scalars(rng, n)
    g1 = np.array([[1, 0, 0, 0, 2, 0, 0, 0]], dtype=np.uint64)
    Pts = np.zeros((n, 8), dtype=np.uint64)
    _lib.check(_lib.load().zk_fixed_base_g1(_lib.ptr(g1), _lib.ptr(K), n, _lib.ptr(Pts)))
    assert np.array_equal(msm_g1(S, Pts), co.g1_msm_bucket_arr(S, Pts, 15))


@pytest.mark.parametrize("n", [5, 4096, 5000])
def test_fixed_base_batches_bit_exact(n):
    """zk_fixed_base_g1/g2 (setup.py:18-69, srs.py:77-85): below 4096 scalars one double-and-add per thread, from
    4096 on the byte-window table kernel -- both against the oracle's k*P, with edge scalars and a non-generator base."""
    rng = np.random.default_rng(900 + n)
    K = rand_fr_limbs(rng, n)
    K[0], K[1], K[2] = 0, limb_row(1), limb_row(o.R - 1)
    K[3] = limb_row(255 << 248 >> 2)            # only the top byte-window set
    K[4] = limb_row((1 << 256) - 1)             # not reduced mod r: the kernel takes all 256 bits, like bn128.multiply
    lib = _lib.load()
    base1 = co.g1_to_arr([co.g1_mul(o.G1, 7)])
    out1 = np.zeros((n, 8), dtype=np.uint64)
    _lib.check(lib.zk_fixed_base_g1(_lib.ptr(base1), _lib.ptr(K), n, _lib.ptr(out1)))
    idx = list(range(8)) + [n // 2, n - 1] if n > 8 else list(range(n))
    assert np.array_equal(out1[idx], co.g1_fixed_base_arr(co.g1_from_arr(base1)[0], K[idx]))
    base2 = co.g2_to_arr([o.G2])
    out2 = np.zeros((n, 16), dtype=np.uint64)
    _lib.check(lib.zk_fixed_base_g2(_lib.ptr(base2), _lib.ptr(K), n, _lib.ptr(out2)))
    for i in idx[:6] + idx[-1:]:
        want = np.zeros(16, dtype=np.uint64)
        co.lib().orc_g2_mul(co._p(base2), co._p(K[i:i + 1].copy()), co._p(want))
        assert np.array_equal(out2[i], want), i
    # infinity base -> all results infinity (zeros)
    zero = np.zeros((1, 8), dtype=np.uint64)
    _lib.check(lib.zk_fixed_base_g1(_lib.ptr(zero), _lib.ptr(K), n, _lib.ptr(out1)))
    assert not out1.any()


@pytest.mark.parametrize("n", [1, 7, 4095, 4096, 6001])
def test_fixed_base_device_buffers_equal_host_buffers_and_oracle(n):
    """zk_fixed_base_g1_dev / _g2_dev (key generation at scale: exponents and points stay in HBM; setup.py:18-69, srs.py:77-85):
    the same points as the host-buffer batch -- both code paths (one double-and-add per thread below 4096 scalars, the byte-window
    table from there on) -- and the oracle's k * P on a few of them, with edge exponents 0, 1, r - 1 and one above r."""
    import torch
    rng = np.random.default_rng(4000 + n)
    K = rand_fr_limbs(rng, n)
    for j, v in enumerate([0, 1, o.R - 1, (1 << 256) - 1][:n]):
        K[j] = limb_row(v)
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    dK = torch.from_numpy(K.view(np.int64)).cuda()
    idx = sorted({0, min(1, n - 1), min(2, n - 1), min(3, n - 1), n // 2, n - 1})
    base1 = co.g1_to_arr([co.g1_mul(o.G1, 11)])
    host1, dev1 = np.zeros((n, 8), dtype=np.uint64), torch.zeros((n, 8), dtype=torch.int64, device="cuda")
    _lib.check(lib.zk_fixed_base_g1(_lib.ptr(base1), _lib.ptr(K), n, _lib.ptr(host1)))
    _lib.check(lib.zk_fixed_base_g1_dev(_lib.ptr(base1), dK.data_ptr(), n, dev1.data_ptr(), st))
    got1 = dev1.cpu().numpy().view(np.uint64)
    assert np.array_equal(got1, host1)
    assert np.array_equal(got1[idx], co.g1_fixed_base_arr(co.g1_from_arr(base1)[0], K[idx]))
    base2 = co.g2_to_arr([o.G2])
    host2, dev2 = np.zeros((n, 16), dtype=np.uint64), torch.zeros((n, 16), dtype=torch.int64, device="cuda")
    _lib.check(lib.zk_fixed_base_g2(_lib.ptr(base2), _lib.ptr(K), n, _lib.ptr(host2)))
    _lib.check(lib.zk_fixed_base_g2_dev(_lib.ptr(base2), dK.data_ptr(), n, dev2.data_ptr(), st))
    got2 = dev2.cpu().numpy().view(np.uint64)
    assert np.array_equal(got2, host2)
    assert np.array_equal(got2[idx], co.g2_fixed_base_arr(o.G2, K[idx]))
    assert lib.zk_fixed_base_g1_dev(None, dK.data_ptr(), n, dev1.data_ptr(), st) == _lib.ZK_ERR_INVALID
    assert lib.zk_fixed_base_g1_dev(_lib.ptr(base1), dK.data_ptr(), 0, dev1.data_ptr(), st) == 0      # n = 0: nothing to do


def test_chunked_msm_small_chunks():
    """MSMs beyond the chunk size (2^24 points; here 2^12 through the test knob) run as consecutive chunks in the
    plan's lanes with the partial sums added on the host: blocking, pipelined and partial forms, bit-exact."""
    import torch
    rng = np.random.default_rng(31)
    n = 5 * 4096 + 123
    S = rand_fr_limbs(rng, n)
    Pts, _ = rand_g1_limbs(rng, n)
    want = co.g1_msm_bucket_arr(S, Pts, 12)
    dS, dP = torch.from_numpy(S.view(np.int64)).cuda(), torch.from_numpy(Pts.view(np.int64)).cuda()
    st = torch.cuda.current_stream().cuda_stream
    plan = MsmPlan(_lib.GROUP_G1, n, chunk_log=12)
    got = plan.run_limbs(dS.data_ptr(), dP.data_ptr(), n, st)[0]
    assert np.array_equal(got, want)
    t = plan.submit(dS.data_ptr(), dP.data_ptr(), n, st)
    with pytest.raises(_lib.ZkhipError):
        plan.submit(dS.data_ptr(), dP.data_ptr(), 100, st)       # nothing else while a chunked MSM is outstanding
    assert np.array_equal(plan.collect_limbs(t)[0], want)
    # a chunked MSM submitted while earlier submissions are in flight runs in the lanes that are free (a prover's merged query
    # behind its other queries); the earlier tickets stay collectable in any order
    want_a, want_b = co.g1_msm_arr(S[:3000], Pts[:3000]), co.g1_msm_arr(S[100:2100], Pts[100:2100])
    ta = plan.submit(dS.data_ptr(), dP.data_ptr(), 3000, st)
    tb = plan.submit(dS.data_ptr() + 32 * 100, dP.data_ptr() + 64 * 100, 2000, st)
    tc = plan.submit(dS.data_ptr(), dP.data_ptr(), n, st)          # one free lane: the chunks take turns in it
    assert np.array_equal(plan.collect_limbs(ta)[0], want_a)
    assert np.array_equal(plan.collect_limbs(tc)[0], want)
    assert np.array_equal(plan.collect_limbs(tb)[0], want_b)
    # lanes are taken wherever one is free: A outstanding, a chunked MSM collected, and the next submission must not be refused
    # because the rotation happens to point at A's lane (it was, with two lanes free, until round 4)
    ta = plan.submit(dS.data_ptr(), dP.data_ptr(), 3000, st)
    tc = plan.submit(dS.data_ptr(), dP.data_ptr(), n, st)
    assert np.array_equal(plan.collect_limbs(tc)[0], want)
    tb = plan.submit(dS.data_ptr() + 32 * 100, dP.data_ptr() + 64 * 100, 2000, st)
    td = plan.submit(dS.data_ptr(), dP.data_ptr(), 3000, st)       # A, B and D outstanding: every lane taken
    with pytest.raises(_lib.ZkhipError):
        plan.submit(dS.data_ptr(), dP.data_ptr(), 100, st)
    assert np.array_equal(plan.collect_limbs(tb)[0], want_b)
    te = plan.submit(dS.data_ptr() + 32 * 100, dP.data_ptr() + 64 * 100, 2000, st)   # B's lane again, A and D still out
    assert np.array_equal(plan.collect_limbs(ta)[0], want_a)
    assert np.array_equal(plan.collect_limbs(te)[0], want_b)
    assert np.array_equal(plan.collect_limbs(td)[0], want_a)
    t3 = [plan.submit(dS.data_ptr(), dP.data_ptr(), 3000, st) for _ in range(3)]
    with pytest.raises(_lib.ZkhipError):
        plan.submit(dS.data_ptr(), dP.data_ptr(), n, st)           # no lane free at all
    for t in t3:
        assert np.array_equal(plan.collect_limbs(t)[0], want_a)
    part = plan.run_partial(dS.data_ptr(), dP.data_ptr(), n, st)
    assert np.array_equal(_lib.limbs_to_ints(np.array(co.g1_to_arr([fold_partials(_lib.GROUP_G1, part)]))), _lib.limbs_to_ints(want.reshape(1, 8)))
    small = plan.run_limbs(dS.data_ptr(), dP.data_ptr(), 3000, st)[0]  # below the chunk size: the ordinary path
    assert np.array_equal(small, co.g1_msm_arr(S[:3000], Pts[:3000]))
    plan.close()


def test_g1_msm_2pow24_chunked_closed_form():
    """BASELINE.json's 2^24 size on one GPU: four 2^22-point chunks through the plan's lanes; P_i = (k0 + i d) G so that the
    result has the closed form (sum s_i (k0 + i d)) G; a few bases are spot-checked against the oracle."""
    import torch
    from zkhip.synthetic import random_scalars
    from helpers import arithmetic_dot, arithmetic_g1_points
    n = 1 << 24
    k0, d = 0x1234567890ABCDEF >> 1, 0x9E3779B1
    lib = _lib.load()
    Pts = arithmetic_g1_points(lib, n, k0, d)
    for i in (0, 1, n // 3, n - 1):
        assert np.array_equal(Pts[i], co.g1_to_arr([co.g1_mul(o.G1, k0 + i * d)])[0])
    S = random_scalars(np.random.default_rng(24), n)
    dS, dP = torch.from_numpy(S.view(np.int64)).cuda(), torch.from_numpy(Pts.view(np.int64)).cuda()
    plan = MsmPlan(_lib.GROUP_G1, n)
    got = plan.run(dS.data_ptr(), dP.data_ptr(), n, torch.cuda.current_stream().cuda_stream)
    assert got == o_point(co.g1_mul(o.G1, arithmetic_dot(S, k0, d)))
    plan.close()


def test_g1_msm_2pow26_one_gpu_closed_form():
    """BASELINE.json's largest size (configs[4] is this MSM sharded over eight GPUs) on ONE GPU: sixteen 2^22-point chunks
    through the plan's lanes, ~7 GB of inputs.  Scalars are drawn on the device (uniform below r), bases are
    P_i = (k0 + i d) G1, the result must be (sum s_i (k0 + i d)) G1 computed by the oracle; a few bases and the first
    scalars' canonicity are spot-checked on the host.  The eight-rank split of the same MSM is a sum of such partials:
    the second half checks that two half-size partial sums fold to the same point (zk_msm_fold_partials), and then that the
    eight 2^23-point chunks of the eight-rank split (shard_range) fold, in rank order, to it as well."""
    import torch
    from zkhip.distributed import fold_partials
    from zkhip.synthetic import ARITH_D, ARITH_K0, arithmetic_dot_device, arithmetic_points, random_scalars_device
    n = 1 << 26
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    dS = random_scalars_device(n, dev, 26)
    head = _lib.limbs_to_ints(dS[:4096].cpu().numpy().view(np.uint64))
    assert max(head) < o.R and len(set(head)) == len(head)
    Pts = arithmetic_points(lib, n)
    for i in (0, 1, n // 3, n - 1):
        assert np.array_equal(Pts[i], co.g1_to_arr([co.g1_mul(o.G1, ARITH_K0 + i * ARITH_D)])[0])
    dP = torch.from_numpy(Pts.view(np.int64)).to(dev)
    del Pts
    st = torch.cuda.current_stream().cuda_stream
    plan = MsmPlan(_lib.GROUP_G1, n)
    want = o_point(co.g1_mul(o.G1, arithmetic_dot_device(dS)))
    assert plan.run(dS.data_ptr(), dP.data_ptr(), n, st) == want
    half = n // 2
    parts = np.stack([plan.run_partial(dS[:half].data_ptr(), dP[:half].data_ptr(), half, st),
                      plan.run_partial(dS[half:].data_ptr(), dP[half:].data_ptr(), half, st)])
    assert fold_partials(_lib.GROUP_G1, parts) == want
    # configs[4] itself, minus the wire: the eight ranks' chunks exactly as zkhip.distributed cuts them (shard_range: contiguous,
    # 2^23 points each), every chunk down to its 128-byte XYZZ partial, the eight partials folded in rank order -- what every rank
    # does with the all-gathered partials (the all-gather is covered by the world-size-2 / 8 gloo tests and the 4-rank rehearsal)
    from zkhip.distributed import shard_range
    parts8 = []
    for rank in range(8):
        lo, hi = shard_range(n, rank, 8)
        assert hi - lo == 1 << 23
        parts8.append(plan.run_partial(dS[lo:hi].data_ptr(), dP[lo:hi].data_ptr(), hi - lo, st))
    assert fold_partials(_lib.GROUP_G1, np.stack(parts8)) == want
    plan.close()


def o_point(pt):
    from zkhip.field import FQ
    return None if pt is None else (FQ(pt[0]), FQ(pt[1]))


@pytest.mark.parametrize("group", ["g1", "g2"])
def test_bound_bases_mode_equals_unbound(group):
    """zk_msm_plan_bind_points: the 13-row table of 2^(20 w) * P_i and one window of 2^19 buckets must give the very same
    points as the ordinary 16-window path -- uniform and skewed scalars, prefixes of the bound bases, infinity among the
    bases, three submissions in flight, and a chunked call (test knob) that walks the table with an offset."""
    import torch
    rng = np.random.default_rng(41)
    g2 = group == "g2"
    n = (1 << 18) + 5000 if not g2 else 9000
    if g2:
        base, _ = rand_g2_limbs(rng, 32)
        Pts = base[rng.integers(0, 32, size=n)]
    else:
        from helpers import arithmetic_g1_points
        Pts = arithmetic_g1_points(_lib.load(), n, 0x1234567890ABCDEF >> 1, 0x9E3779B1)
    Pts[7] = 0                                                        # an infinity base
    from zkhip.synthetic import random_scalars
    S = random_scalars(rng, n)
    S[3] = 0
    S[4] = limb_row(o.R - 1)
    W = S.copy()
    pick = rng.random(n)
    W[pick < 0.3] = limb_row(1)
    W[(pick >= 0.3) & (pick < 0.5)] = 0
    dP = torch.from_numpy(Pts.view(np.int64)).cuda()
    dS, dW = torch.from_numpy(S.view(np.int64)).cuda(), torch.from_numpy(W.view(np.int64)).cuda()
    st = torch.cuda.current_stream().cuda_stream
    plan = MsmPlan(_lib.GROUP_G2 if g2 else _lib.GROUP_G1, max(n, (1 << 17) + 1), chunk_log=18)   # chunking at testable sizes
    want_s = plan.run_limbs(dS.data_ptr(), dP.data_ptr(), n, st)
    want_w = plan.run_limbs(dW.data_ptr(), dP.data_ptr(), n, st)
    want_pre = plan.run_limbs(dS.data_ptr(), dP.data_ptr(), 5001, st)
    with pytest.raises(_lib.ZkhipError):
        plan.run_limbs(dS.data_ptr(), None, n, st)                    # nothing bound yet
    plan.bind(dP.data_ptr(), n, st)
    for _ in range(2):
        got = plan.run_limbs(dS.data_ptr(), None, n, st)
        assert np.array_equal(got[0], want_s[0]) and got[1] == want_s[1]
    got = plan.run_limbs(dW.data_ptr(), None, n, st)
    assert np.array_equal(got[0], want_w[0])
    got = plan.run_limbs(dS.data_ptr(), None, 5001, st)               # a prefix of the bound bases
    assert np.array_equal(got[0], want_pre[0])
    if n <= (1 << 18):                                                # unchunked sizes: keep three in flight, mixed with unbound calls
        t = [plan.submit(dS.data_ptr(), None, n, st), plan.submit(dW.data_ptr(), dP.data_ptr(), n, st), plan.submit(dW.data_ptr(), None, n, st)]
        r = [plan.collect_limbs(x) for x in t]
        assert np.array_equal(r[0][0], want_s[0]) and np.array_equal(r[1][0], want_w[0]) and np.array_equal(r[2][0], want_w[0])
    with pytest.raises(_lib.ZkhipError):
        plan.run_limbs(dS.data_ptr(), None, n + 1, st)                # more than bound
    plan.bind(None, 0, st)                                            # unbind
    with pytest.raises(_lib.ZkhipError):
        plan.run_limbs(dS.data_ptr(), None, n, st)
    plan.close()


def test_g1_top_window_spreading_edges():
    """Above 2^17 points (16-bit windows) and in bound-bases mode a G1 scalar k >= 2^240 is run as k + m*r (the top window /
    table row would otherwise use a fraction of its buckets).  The values around the 2^240 threshold and just below r sit
    at odd and even positions here; both modes must give the closed form (sum s_i k_i) * G1."""
    import torch
    from zkhip.synthetic import random_scalars
    from helpers import arithmetic_dot, arithmetic_g1_points
    n = (1 << 17) + 4097
    k0, d = 0x0123456789ABCDE, 0x9E3779B1
    rng = np.random.default_rng(240)
    S = random_scalars(rng, n)
    edge = [(1 << 240) - 1, 1 << 240, (1 << 240) + 1, o.R - 1, o.R - 2, 1 << 253, (1 << 253) + (1 << 240), o.R - (1 << 240),
            (1 << 241) - 1, 0, 1, o.R - (1 << 239)]
    for j, v in enumerate(edge):
        S[1000 + 2 * j] = limb_row(v)          # even positions
        S[2001 + 2 * j] = limb_row(v)          # odd positions: the ones the 16-window path shifts by r
        S[n - 1 - j] = limb_row(v)
    Pts = arithmetic_g1_points(_lib.load(), n, k0, d)
    want = co.g1_mul(o.G1, arithmetic_dot(S, k0, d))
    dS, dP = torch.from_numpy(S.view(np.int64)).cuda(), torch.from_numpy(Pts.view(np.int64)).cuda()
    st = torch.cuda.current_stream().cuda_stream
    ints = lambda pt: tuple(int(c) for c in pt)
    plan = MsmPlan(_lib.GROUP_G1, n)
    assert ints(plan.run(dS.data_ptr(), dP.data_ptr(), n, st)) == ints(want)
    plan.bind(dP.data_ptr(), n, st)
    assert ints(plan.run(dS.data_ptr(), None, n, st)) == ints(want)
    # the same scalars on a short prefix go through the 15-bit-window path, which leaves them alone
    m = 3000
    assert ints(plan.run(dS.data_ptr(), dP.data_ptr(), m, st)) == ints(co.g1_mul(o.G1, arithmetic_dot(S[:m], k0, d)))
    plan.close()


def test_device_scalars_at_or_above_2pow255_fail_loudly():
    """_dev entry points cannot inspect device scalars on the host.  Every canonical scalar is handled exactly, and so is any
    value up to 2^254 - 2^240 (k*P for k >= r is the same point as (k mod r)*P); a larger one whose signed digits do not fit
    the windows (anything from 2^255; from about 2^254 with 15-bit windows) would lose 2^(W c) * P, so the prepare kernel flags it and
    the call that collects the submission fails with ZK_ERR_INVALID -- blocking and pipelined forms alike -- and the plan
    stays usable afterwards.  Never a wrong point with ZK_OK."""
    import torch
    rng = np.random.default_rng(255)
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    for n in (300, 5000, (1 << 17) + 77):                            # 8-, 15- and 16-bit windows
        S = rand_fr_limbs(rng, n)
        Pts = arithmetic_g1_points(lib, n, 12345, 7)
        S[n // 2] = limb_row(o.R + 5)                                 # non-canonical but < 2^255: still exact
        S[n // 3] = limb_row(o.R + (1 << 250))
        dS, dP = torch.from_numpy(S.view(np.int64)).cuda(), torch.from_numpy(Pts.view(np.int64)).cuda()
        plan = MsmPlan(_lib.GROUP_G1, n)
        ks = [12345 + 7 * i for i in range(n)]
        sv = _lib.limbs_to_ints(S)
        want = co.g1_mul(o.G1, sum(s * k for s, k in zip(sv, ks)) % o.R)
        got = plan.run(dS.data_ptr(), dP.data_ptr(), n, st)
        assert (int(got[0]), int(got[1])) == want
        for v in (1 << 255, (1 << 255) - 19, (1 << 256) - 1):
            bad = S.copy()
            bad[7] = limb_row(v)
            dB = torch.from_numpy(bad.view(np.int64)).cuda()
            with pytest.raises(_lib.ZkhipError) as e:
                plan.run(dB.data_ptr(), dP.data_ptr(), n, st)
            assert e.value.code == _lib.ZK_ERR_INVALID
        t_ok, t_bad = plan.submit(dS.data_ptr(), dP.data_ptr(), n, st), plan.submit(dB.data_ptr(), dP.data_ptr(), n, st)
        with pytest.raises(_lib.ZkhipError):
            plan.collect_limbs(t_bad)
        ok = limbs_to_pt(plan.collect_limbs(t_ok))
        assert ok == want
        got = plan.run(dS.data_ptr(), dP.data_ptr(), n, st)           # the plan is not poisoned
        assert (int(got[0]), int(got[1])) == want
        plan.close()


def limbs_to_pt(res):
    from zkhip.field import limbs_to_g1
    limbs, inf = res
    if inf:
        return None
    p = limbs_to_g1(limbs)[0]
    return (int(p[0]), int(p[1]))


def test_chunked_submission_with_a_bad_chunk_leaves_the_plan_usable():
    import torch
    rng = np.random.default_rng(256)
    n = 3 * 4096 + 50
    lib = _lib.load()
    S = rand_fr_limbs(rng, n)
    Pts = arithmetic_g1_points(lib, n, 999, 3)
    bad = S.copy()
    bad[2 * 4096 + 5] = limb_row((1 << 256) - 1)
    dS, dB, dP = (torch.from_numpy(x.view(np.int64)).cuda() for x in (S, bad, Pts))
    st = torch.cuda.current_stream().cuda_stream
    plan = MsmPlan(_lib.GROUP_G1, n, chunk_log=12)
    want = co.g1_mul(o.G1, sum(s * (999 + 3 * i) for i, s in enumerate(_lib.limbs_to_ints(S))) % o.R)
    with pytest.raises(_lib.ZkhipError):
        plan.run(dB.data_ptr(), dP.data_ptr(), n, st)
    got = plan.run(dS.data_ptr(), dP.data_ptr(), n, st)
    assert (int(got[0]), int(got[1])) == want
    t = plan.submit(dS.data_ptr(), dP.data_ptr(), 1000, st)           # and ordinary submissions work again
    assert limbs_to_pt(plan.collect_limbs(t)) == co.g1_mul(o.G1, sum(s * (999 + 3 * i) for i, s in enumerate(_lib.limbs_to_ints(S[:1000]))) % o.R)
    plan.close()


def test_invalid_host_scalars_for_device_entries():
    """coset_shift / zinv are host-side arguments of device entry points: zero or non-canonical values are refused."""
    import torch
    from zkhip.device import NttPlan, fr_quotient
    d = torch.zeros((16, 4), dtype=torch.int64, device="cuda")
    plan = NttPlan(4)
    st = torch.cuda.current_stream().cuda_stream
    for bad in (0, o.R, o.R + 5):
        with pytest.raises(_lib.ZkhipError) as e:
            plan.run(d.data_ptr(), True, bad, st)
        assert e.value.code == _lib.ZK_ERR_INVALID
    plan.run(d.data_ptr(), True, 5, st)
    with pytest.raises(_lib.ZkhipError):
        fr_quotient(d.data_ptr(), d.data_ptr(), d.data_ptr(), d.data_ptr(), o.R, 16, st)
    arr = _lib.ints_to_limbs([1, 2, 3, 4])
    k = _lib.ints_to_limbs([0])
    assert _lib.load().zk_ntt_fr(_lib.ptr(arr), 2, 1, _lib.ptr(k)) == _lib.ZK_ERR_INVALID
    torch.cuda.synchronize()
