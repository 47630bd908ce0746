"""CPU tests of bench.py's own launcher (`python bench.py --gpus N` with no rank environment) and of the
synthetic-workload helpers it uses.  No GPU: the launcher decides before any HIP call."""
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
BENCH = os.path.join(ROOT, "bench.py")


def _run(*argv, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + list(argv), capture_output=True, text=True, env=env, timeout=300)


def test_gpus_2_takes_the_launcher_path():
    """--gpus 2 without WORLD_SIZE must go through launch_ranks: here (no device) it refuses with exit code 2 and names the
    device count -- a run that ignored --gpus would instead die on 'needs a HIP device' from the single-rank path."""
    r = _run("--gpus", "2", "--steps", "2", "--warmup", "1")
    assert r.returncode == 2
    assert "needs 2 HIP devices, 0 visible" in r.stderr


def test_dry_launch_prints_one_rank_per_gpu_command():
    r = _run("--gpus", "4", "--steps", "3", "--warmup", "2", "--dry-launch")
    assert r.returncode == 0, r.stderr
    rec = json.loads(r.stdout.strip().splitlines()[-1])
    cmd = rec["launch"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert os.path.basename(cmd[cmd.index("--master-port") + 2]) == "bench.py"
    tail = cmd[cmd.index("--master-port") + 3:]
    assert tail[:6] == ["--gpus", "4", "--steps", "3", "--warmup", "2"]


def test_rank_environment_skips_the_launcher_and_checks_world_size():
    """Started by the driver's torch.distributed.run the script is a rank, not a launcher; a world size that disagrees
    with --gpus is an error (before any GPU call)."""
    r = _run("--gpus", "2", env_extra={"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0
    assert "--gpus 2 but the launcher started 4 ranks" in (r.stderr + r.stdout)


def test_single_gpu_without_device_fails_loudly():
    r = _run("--steps", "1", "--warmup", "0")
    assert r.returncode != 0
    assert "needs a HIP device" in (r.stderr + r.stdout)


def test_arithmetic_dot_host_and_device_agree_with_big_integers():
    import torch
    sys.path.insert(0, os.path.join(ROOT, "interactive-zkp-study_amd"))
    from zkhip import _lib, synthetic as sy
    rng = np.random.default_rng(11)
    s = sy.random_scalars(rng, 70000)
    ints = _lib.limbs_to_ints(s)
    assert max(ints) < sy.R_MOD
    for first in (0, (1 << 25) + 7, (1 << 29) + 12345):
        want = sum(v * (sy.ARITH_K0 + (first + i) * sy.ARITH_D) for i, v in enumerate(ints)) % sy.R_MOD
        assert sy.arithmetic_dot(s, first=first) == want
        assert sy.arithmetic_dot_device(torch.from_numpy(s.view(np.int64)), first=first) == want
    d = sy.random_scalars_device(50000, "cpu", 5)
    vals = _lib.limbs_to_ints(d.numpy().view(np.uint64))
    assert max(vals) < sy.R_MOD and max(vals).bit_length() >= 252 and len(set(vals)) == len(vals)


def test_committed_issue_rate_reads_the_profile_or_returns_none(tmp_path):
    """roofline.alu.valu_issue_pmc quotes profiles/*_pmc_sq_summary.csv (tools/collect_profiles.py); a kernel that is not in
    the file, or no file at all, gives None instead of a made-up figure."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", BENCH)
    mod = importlib.util.module_from_spec(spec)
    argv = sys.argv
    sys.argv = ["bench.py"]
    try:
        spec.loader.exec_module(mod)
    finally:
        sys.argv = argv
    got = mod.committed_issue_rate("msm_accumulate_kernel<zk::Fe<zk::FpTag> >")
    if got is None:
        # the committed counter pass belongs to other arithmetic sources (field.h / curve.h / the accumulate kernel changed since it
        # was taken): not quoting it IS the guard working; a missing tag counts as stale too
        import glob, json
        metas = sorted(glob.glob(os.path.join(mod.ROOT, "profiles", "*_pmc_sq_summary.meta.json")))
        tagged = json.load(open(metas[-1])).get("arithmetic_source_sha256") if metas else None
        assert tagged != mod.arithmetic_source_hash()
    else:
        assert got["peak"] == 0.25 and 0.0 < got["frac"] <= 1.0 and got["source"].startswith("profiles/")
        assert abs(got["frac"] - got["insts_per_simd_cycle"] / 0.25) < 1e-12
        assert got["arithmetic_source_sha256"] == mod.arithmetic_source_hash()
    assert mod.committed_issue_rate("no_such_kernel") is None
    root = mod.ROOT
    mod.ROOT = str(tmp_path)
    try:
        assert mod.committed_issue_rate("msm_accumulate_kernel<zk::Fe<zk::FpTag> >") is None
    finally:
        mod.ROOT = root


def _load_bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod2", BENCH)
    mod = importlib.util.module_from_spec(spec)
    argv = sys.argv
    sys.argv = ["bench.py"]
    try:
        spec.loader.exec_module(mod)
    finally:
        sys.argv = argv
    return mod


def test_python_baseline_fit_and_labelled_extrapolation():
    """Row D4: the pure-Python baseline is timed at small sizes, fitted (linear for the MSM, n log n for the fft) and extrapolated
    to the config sizes, labelled as extrapolated; the host's CPU model / nproc travel with it."""
    mod = _load_bench()
    sizes = [64, 128, 256, 512, 1024]
    lin = mod.fit_and_extrapolate(sizes, [3.1e-3 * n for n in sizes], (20, 24, 26), False)
    assert abs(lin["a_seconds"] - 3.1e-3) < 1e-12 and lin["max_rel_residual"] < 1e-9
    assert abs(lin["extrapolated_seconds"]["2^20"] - 3.1e-3 * (1 << 20)) < 1e-6
    assert set(lin["extrapolated_seconds"]) == {"2^20", "2^24", "2^26"} and "EXTRAPOLATED" in lin["label"]
    nl = mod.fit_and_extrapolate([256, 1024, 4096], [2e-6 * n * np.log2(n) for n in (256, 1024, 4096)], (22, 24), True)
    assert abs(nl["extrapolated_seconds"]["2^22"] - 2e-6 * (1 << 22) * 22) < 1e-6 and "log2" in nl["model"]
    noisy = mod.fit_and_extrapolate(sizes, [3e-3 * n * (1.1 if i % 2 else 0.9) for i, n in enumerate(sizes)], (20,), False)
    assert 0.05 < noisy["max_rel_residual"] < 0.2
    host = mod.host_description()
    assert host["nproc"] == os.cpu_count() and (host["cpu_model"] is None or isinstance(host["cpu_model"], str))


def test_static_figures_are_quoted_only_for_the_sources_they_came_from(tmp_path):
    """roofline.alu.mad_floor (multiply-adds per addition, read off a code object) and valu_issue_pmc (a committed counter pass) are
    not measured by a bench run: they carry the hash of the arithmetic sources they belong to and vanish when those change."""
    import json
    import shutil
    mod = _load_bench()
    sc = mod.static_counts()
    assert sc is not None and sc["mads_per_madd"] > 1000 and sc["arithmetic_source_sha256"] == mod.arithmetic_source_hash()
    # a copy of the tree with one changed byte in field.h: both figures are dropped
    root = tmp_path / "repo"
    csrc = root / "interactive-zkp-study_amd" / "csrc"
    csrc.mkdir(parents=True)
    for name in ("field.h", "curve.h", "msm_impl.h"):
        shutil.copy(os.path.join(ROOT, "interactive-zkp-study_amd", "csrc", name), csrc / name)
    shutil.copytree(os.path.join(ROOT, "profiles"), root / "profiles")
    # the copy's counter pass is stamped with the copy's sources here, whatever the state of the committed one (which goes stale
    # whenever the arithmetic changes and stays unquoted until tools/collect_profiles.py has been run again on the GPU box)
    import glob
    latest = sorted(glob.glob(str(root / "profiles" / "*_pmc_sq_summary.csv")))[-1]
    json.dump({"arithmetic_source_sha256": mod.arithmetic_source_hash()}, open(latest[:-4] + ".meta.json", "w"))
    real = mod.ROOT
    mod.ROOT = str(root)
    try:
        assert mod.static_counts() is not None
        assert mod.committed_issue_rate("msm_accumulate_kernel<zk::Fe<zk::FpTag> >") is not None
        with open(csrc / "field.h", "a") as f:
            f.write("// changed\n")
        assert mod.static_counts() is None
        assert mod.committed_issue_rate("msm_accumulate_kernel<zk::Fe<zk::FpTag> >") is None
    finally:
        mod.ROOT = real
