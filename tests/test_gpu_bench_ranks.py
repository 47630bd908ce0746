"""GPU tests (-m gpu) of bench.py's multi-rank paths, each as a fresh child process (`python bench.py ...`, which starts its
own ranks before it touches a GPU):

  * `--gpus 4 --rehearse`: the N-rank code of BASELINE.json configs[4] -- ONE 2^26-point MSM sharded over the ranks by contiguous
    chunks, the all-gather of the 128-byte partials and the rank-order fold -- plus ONE 2^24-point NTT spread over the ranks
    (four-step, one all-to-all), four ranks sharing this box's one GPU over gloo.  Four, not eight: a GPU box of this pool admits at
    most six processes on its card (pytest itself is one), and the eight-rank run belongs to the driver's 8-GPU node.  The rank
    count only changes the chunk boundaries (zkhip.distributed.shard_range) and the fold length.
  * `--gpus 1 --force-dist`: the same code over RCCL -- init_process_group("nccl", device_id=...), the side-stream all-gather and
    the ExchangeWorker thread -- on one rank, which is what one GPU can run of it.

Both assert the required keys of the line, so an N > 1 line without `roofline` / `cpu_baseline` fails here."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "config", "roofline", "cpu_baseline")


def _bench(*argv, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, BENCH] + list(argv), capture_output=True, text=True, env=env, timeout=timeout)
    assert r.returncode == 0, "bench.py %s failed (%d):\n%s\n%s" % (" ".join(argv), r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "expected ONE JSON line, got %d:\n%s" % (len(lines), r.stdout[-2000:])
    rec = json.loads(lines[0])
    for key in REQUIRED:
        assert key in rec, key
    return rec


def _check_roofline_and_cpu(rec):
    roof = rec["roofline"]
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert roof["achieved"] > 0 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-12
    cpu = rec["cpu_baseline"]
    assert cpu is not None and cpu["kind"] == "port" and cpu["cores"] == 1 and cpu["value"] > 0
    assert "matches GPU MSM of the same sample: True" in cpu["sample"]
    assert cpu["host"]["nproc"] >= 1
    assert "bit-identical to the GPU result for the same points: True" in cpu["compiled_c"]["sample"]
    fit = cpu["extrapolation"]
    assert fit["every_sample_equals_gpu"] and "EXTRAPOLATED" in fit["label"] and set(fit["extrapolated_seconds"]) == {"2^20", "2^24", "2^26"}


def test_four_rank_rehearsal_of_the_sharded_2pow26_msm_and_the_distributed_ntt():
    rec = _bench("--gpus", "4", "--rehearse", "--steps", "2", "--warmup", "1", "--shard-total-log", "26", "--dist-ntt-log-n", "24",
                 "--cpu-sample", "128", "--ntt-log-n", "18", "--dist-groth16-log-m", "16")
    assert rec["n_gpus"] == 4 and rec["steps"] == 2 and rec["warmup"] == 1 and rec["scaling"] == "weak"
    assert "REHEARSAL" in rec["config"]["collectives"]
    seen = rec["config"]["ranks_seen"]                                   # four ranks, ONE physical device: the line says so itself
    assert seen["ranks"] == 4 and seen["distinct_devices"] == 1 and sorted(r["rank"] for r in seen["by_rank"]) == [0, 1, 2, 3]
    extra = rec["extra"]
    assert extra["verified_closed_form"] is True
    one = extra["sharded_one_msm"]
    assert one["log_n_total"] == 26 and one["points_per_gpu"] == 1 << 24 and one["verified_closed_form"] is True
    dn = extra["dist_ntt"]
    assert dn["log_n"] == 24 and dn["roundtrip_exact"] is True
    dg = extra["dist_groth16"]                                           # the proof with every vector spread over the four ranks
    assert dg["log_m"] == 16 and dg["verified_closed_form"] is True and dg["coefficients_per_rank"] == 1 << 14
    assert extra["ntt"]["roundtrip_exact"] is True and "all_gpus" in extra["ntt"]
    _check_roofline_and_cpu(rec)


def test_one_rank_over_rccl_takes_the_multi_rank_path():
    rec = _bench("--gpus", "1", "--force-dist", "--steps", "5", "--warmup", "2", "--shard-total-log", "22", "--dist-ntt-log-n", "22",
                 "--cpu-sample", "128", "--ntt-log-n", "18", "--groth16-log-m", "0", "--plonk-log-n", "0", "--no-bound", "--no-witness-like",
                 "--dist-groth16-log-m", "18")
    assert rec["n_gpus"] == 1 and rec["steps"] == 5
    assert rec["config"]["collectives"].startswith("RCCL")
    assert rec["config"]["ranks_seen"]["ranks"] == 1 and rec["config"]["ranks_seen"]["distinct_devices"] == 1
    extra = rec["extra"]
    assert extra["verified_closed_form"] is True
    assert extra["sharded_one_msm"]["verified_closed_form"] is True
    assert extra["dist_ntt"]["roundtrip_exact"] is True
    assert extra["dist_groth16"]["verified_closed_form"] is True and extra["dist_groth16"]["coefficients_per_rank"] == 1 << 18
    _check_roofline_and_cpu(rec)
