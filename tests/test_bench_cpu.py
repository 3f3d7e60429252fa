"""bench.py's launcher path without a GPU: `--gpus N` spawns its own ranks on 127.0.0.1
(gloo self-test workload: barrier + max-over-ranks timing, no kernel) and refuses
inconsistent launches loudly."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=300)


def test_gpus_2_spawns_two_ranks_over_gloo():
    r = _run(["--gpus", "2", "--workload", "launcher-selftest", "--steps", "4", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                      # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 4 and out["warmup"] == 1
    # max over ranks: rank 1 sleeps 4 ms per step, rank 0 only 2 ms
    assert out["ms_per_step"] >= 3.9


def test_world_size_mismatch_is_an_error():
    r = _run(["--gpus", "2"], {"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=4" in r.stderr


def test_more_ranks_than_gpus_is_an_error():
    import torch
    have = torch.cuda.device_count()
    r = _run(["--gpus", str(have + 1), "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0 and "visible" in r.stderr


@pytest.mark.gpu
def test_gpus_2_on_a_one_gpu_box_fails_loudly():
    import torch
    if torch.cuda.device_count() != 1:
        pytest.skip("needs exactly one visible GPU")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0 and "only 1 GPU" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_argument_parsing_of_the_workloads():
    """every workload / shape / dtype / backend the driver (or a test) may ask for parses; nonsense does not"""
    sys.path.insert(0, ROOT)
    import bench
    ap = bench.parser()
    a = ap.parse_args([])
    assert (a.gpus, a.workload, a.shape, a.dtype, a.backend, a.share_gpu, a.rows) == (1, "fista", "c4", "bf16", "nccl", False, None)
    a = ap.parse_args("--gpus 8 --steps 7 --warmup 2 --workload c3 --dtype f32".split())
    assert (a.gpus, a.steps, a.warmup, a.workload, a.dtype) == (8, 7, 2, "c3", "f32")
    a = ap.parse_args("--workload em --shape c5 --rows 8192 --backend gloo --share-gpu".split())
    assert (a.workload, a.shape, a.rows, a.backend, a.share_gpu) == ("em", "c5", 8192, "gloo", True)
    for bad in (["--workload", "c9"], ["--shape", "c2"], ["--dtype", "fp8"], ["--backend", "mpi"]):
        with pytest.raises(SystemExit):
            ap.parse_args(bad)
    assert bench.EM_SHAPES["c5"][1:4] == (64, 256, 0.1) and bench.EM_SHAPES["c4"][1:3] == (256, 1024)


def test_share_gpu_needs_gloo():
    r = _run(["--gpus", "2", "--share-gpu", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0 and "--share-gpu needs --backend gloo" in r.stderr
