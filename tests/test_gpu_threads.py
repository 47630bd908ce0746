"""Several host threads in the library at once (plans are per thread; the library's process-wide state --
the pool of MSM streams (csrc/msm.h lane_stream_next), the per-thread plan caches behind the host-buffer calls -- has to hold up
under concurrent callers).  ctypes releases the GIL for the duration of a call, so the calls below really overlap.  Every result is
compared with the oracle's, computed beforehand on the main thread."""
import ctypes
import threading

import numpy as np
import pytest
import torch

import c_oracle as co
import py_ref as o
from helpers import rand_fr_limbs, rand_g1_limbs, rand_g2_limbs
from zkhip import _lib
from zkhip.device import MsmPlan, NttPlan

pytestmark = pytest.mark.gpu


def _host_msm_g1(S, P):
    out = np.zeros(8, dtype=np.uint64)
    inf = ctypes.c_int(-1)
    rc = _lib.load().zk_msm_g1(_lib.ptr(S), _lib.ptr(P), S.shape[0], _lib.ptr(out), ctypes.byref(inf))
    assert rc == 0, _lib.load().zk_last_error()
    return out


def _host_msm_g2(S, P):
    out = np.zeros(16, dtype=np.uint64)
    inf = ctypes.c_int(-1)
    rc = _lib.load().zk_msm_g2(_lib.ptr(S), _lib.ptr(P), S.shape[0], _lib.ptr(out), ctypes.byref(inf))
    assert rc == 0, _lib.load().zk_last_error()
    return out


def _host_ntt(vals, log_n):
    a = vals.copy()
    rc = _lib.load().zk_ntt_fr(_lib.ptr(a), log_n, 0, None)
    assert rc == 0, _lib.load().zk_last_error()
    return a


def test_four_threads_host_buffer_calls():
    """Four threads, each with inputs of its own: G1 MSM (two sizes either side of the window-width switch), G2 MSM and a transform
    through the host-buffer entry points (the per-thread plan caches), five rounds each."""
    jobs = []
    for t in range(4):
        rng = np.random.default_rng(9100 + t)
        n1, n2, ng2, log_n = 300 + 17 * t, 1500 + 111 * t, 60 + t, 9 + t
        S1, (P1, _) = rand_fr_limbs(rng, n1), rand_g1_limbs(rng, n1)
        S2, (P2, _) = rand_fr_limbs(rng, n2), rand_g1_limbs(rng, n2)
        Sg, (Pg, _) = rand_fr_limbs(rng, ng2), rand_g2_limbs(rng, ng2)
        V = rand_fr_limbs(rng, 1 << log_n)
        want = (co.g1_msm_arr(S1, P1), co.g1_msm_arr(S2, P2), co.g2_msm_arr(Sg, Pg),
                co.ntt_arr(V.copy(), o.get_root_of_unity(1 << log_n), False))
        jobs.append(((S1, P1), (S2, P2), (Sg, Pg), (V, log_n), want))
    errors = []
    start = threading.Barrier(4)

    def work(t):
        try:
            (S1, P1), (S2, P2), (Sg, Pg), (V, log_n), want = jobs[t]
            start.wait(timeout=300)
            for _ in range(5):
                got = (_host_msm_g1(S1, P1), _host_msm_g1(S2, P2), _host_msm_g2(Sg, Pg), _host_ntt(V, log_n))
                for k, (g, w) in enumerate(zip(got, want)):
                    if not np.array_equal(g, w):
                        errors.append("thread %d result %d differs" % (t, k))
        except Exception as e:  # noqa: BLE001 -- reported on the main thread
            errors.append("thread %d: %r" % (t, e))

    threads = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors


def test_three_threads_plans_of_their_own_pipelined():
    """Three threads, each with a G1 plan of its own and three submissions in flight (streams from the shared pool of three), and a
    fourth running transforms on a stream of its own: results against the oracle."""
    dev = torch.device("cuda", 0)
    n = 20000
    per_thread = []
    for t in range(3):
        rng = np.random.default_rng(9200 + t)
        S = [rand_fr_limbs(rng, n) for _ in range(3)]
        P, _ = rand_g1_limbs(rng, n)
        want = [co.g1_msm_bucket_arr(s, P, 12) for s in S]
        dS = [torch.from_numpy(s.view(np.int64)).to(dev) for s in S]
        dP = torch.from_numpy(P.view(np.int64)).to(dev)
        per_thread.append((dS, dP, want))
    log_n = 14
    rng = np.random.default_rng(9300)
    V = rand_fr_limbs(rng, 1 << log_n)
    want_ntt = co.ntt_arr(V.copy(), o.get_root_of_unity(1 << log_n), False)
    dV0 = torch.from_numpy(V.view(np.int64)).to(dev)
    torch.cuda.synchronize()
    errors = []
    start = threading.Barrier(4)

    def msm_worker(t):
        try:
            torch.cuda.set_device(0)
            dS, dP, want = per_thread[t]
            plan = MsmPlan(_lib.GROUP_G1, n)
            start.wait(timeout=300)
            for _ in range(4):
                tickets = [plan.submit(s.data_ptr(), dP.data_ptr(), n) for s in dS]
                for k, tk in enumerate(tickets):
                    got, inf = plan.collect_limbs(tk)
                    if inf or not np.array_equal(got, want[k]):
                        errors.append("msm thread %d submission %d differs" % (t, k))
            plan.close()
        except Exception as e:  # noqa: BLE001
            errors.append("msm thread %d: %r" % (t, e))

    def ntt_worker():
        try:
            torch.cuda.set_device(0)
            st = torch.cuda.Stream()
            plan = NttPlan(log_n)
            start.wait(timeout=300)
            for _ in range(12):
                with torch.cuda.stream(st):
                    d = dV0.clone()
                    plan.run(d.data_ptr(), stream=st.cuda_stream)
                    st.synchronize()
                    if not np.array_equal(d.cpu().numpy().view(np.uint64), want_ntt):
                        errors.append("transform differs")
            plan.close()
        except Exception as e:  # noqa: BLE001
            errors.append("ntt thread: %r" % (e,))

    threads = [threading.Thread(target=msm_worker, args=(t,)) for t in range(3)] + [threading.Thread(target=ntt_worker)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
