#!/usr/bin/env python3
"""Where a workgroup of the NTT pass kernels spends its life (measurement build only).

    make -C tools nttstamps                  # the library with -DZK_NTT_STAMPS in ntt.hip -> tmp_variants/libzkhip_nttstamps.so
    python3 tools/ntt_phase_probe.py --lib tmp_variants/libzkhip_nttstamps.so [--log-n 22]

Thread 0 of every workgroup stamps the 100 MHz wall clock at its phase boundaries (csrc/ntt.hip NTT_STAMP).  Printed per pass: the
median share of a workgroup's life in each phase, the spread of the workgroups' start times, and for one CU the life lines of the
workgroups it ran."""
import argparse, ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "interactive-zkp-study_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", required=True)
    ap.add_argument("--log-n", type=int, default=22)
    ap.add_argument("--cu-lines", type=int, default=40)
    args = ap.parse_args()
    import torch
    from zkhip import _lib
    _lib.LIB_PATH = args.lib
    from zkhip.device import NttPlan
    lib = _lib.load()
    raw = ctypes.CDLL(os.path.abspath(args.lib))
    n = 1 << args.log_n
    rng = np.random.default_rng(3)
    x = torch.from_numpy(rng.integers(0, 1 << 62, size=(n, 4), dtype=np.int64)).cuda()
    plan = NttPlan(args.log_n)
    stamps = torch.zeros((3, 4096, 16), dtype=torch.int64, device="cuda")
    for _ in range(5):
        plan.run(x.data_ptr())
    torch.cuda.synchronize()
    assert raw.zk_ntt_set_stamp_buffer(ctypes.c_void_p(stamps.data_ptr())) == 0
    plan.run(x.data_ptr())
    torch.cuda.synchronize()
    s = stamps.cpu().numpy().astype(np.int64)
    tick = 0.01   # us per tick of the 100 MHz clock
    for p in range(3):
        rows = s[p][s[p][:, 0] != 0]
        if not len(rows):
            continue
        t0 = rows[:, 0].min()
        nround = int((rows[0, 3:12] != 0).sum())
        last_round = 2 + nround
        life = (rows[:, 13] - rows[:, 0]) * tick
        phases = [("load+convert", rows[:, 1] - rows[:, 0]), ("barrier", rows[:, 2] - rows[:, 1])]
        for r in range(nround):
            phases.append(("round %d (+barrier)" % r, rows[:, 3 + r] - rows[:, 2 + r]))
        phases += [("twiddle+store issue", rows[:, 12] - rows[:, last_round]), ("store drain", rows[:, 13] - rows[:, 12])]
        print("pass %d: %d workgroups, kernel span %.1f us, workgroup life median %.1f us (p10 %.1f, p90 %.1f)" % (
            p, len(rows), (rows[:, 13].max() - t0) * tick, np.median(life), np.percentile(life, 10), np.percentile(life, 90)))
        for name, d in phases:
            d = d * tick
            print("    %-22s median %6.2f us  p10 %6.2f  p90 %6.2f   (%4.1f %% of the life)" % (name, np.median(d), np.percentile(d, 10), np.percentile(d, 90),
                                                                                                 100 * np.median(d) / np.median(life)))
        starts = np.sort((rows[:, 0] - t0) * tick)
        print("    start times: " + " ".join("%.1f" % starts[int(q * (len(starts) - 1))] for q in (0, .1, .2, .24, .26, .3, .4, .5, .6, .7, .8, .9, 1.0)))
        hw = rows[:, 15]
        cu_key = ((hw >> 32) & 0xF) * 1024 + (((hw >> 13) & 7) * 32 + ((hw >> 12) & 1) * 16 + ((hw >> 8) & 0xF))
        keys, counts = np.unique(cu_key, return_counts=True)
        print("    distinct CUs seen %d; workgroups per CU min %d max %d" % (len(keys), counts.min(), counts.max()))
        k = keys[len(keys) // 2]
        mine = rows[cu_key == k]
        mine = mine[np.argsort(mine[:, 0])]
        print("    life lines on one CU (us from the kernel's first stamp): start | loads done | first barrier | rounds... | stores issued | done  simd")
        for r in mine[:args.cu_lines]:
            pts = [r[0], r[1], r[2]] + [r[3 + i] for i in range(nround)] + [r[12], r[13]]
            print("      " + " ".join("%7.2f" % ((v - t0) * tick) for v in pts) + "   simd %d" % ((int(r[15]) >> 4) & 3))
    plan.close()


if __name__ == "__main__":
    main()
