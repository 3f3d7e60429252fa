"""bench.py's MULTI-RANK code paths executed on the one GPU of the test box: two ranks sharing cuda:0, collectives
over gloo (`--backend gloo --share-gpu`).  RCCL wants one GPU per rank, so the driver's 8-GPU run is the first time
the nccl backend carries these lines -- everything else of the N > 1 path (sharding, the all-reduced objective, the
global stop rule of the time-to-tol leg, the one M-step message per EM step, the all-reduced line-search decisions)
runs here.  Figures of such a run are code-path evidence, never performance."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _bench(args, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                      # ONE JSON line, from rank 0
    return json.loads(lines[0])


SHARED = ["--gpus", "2", "--backend", "gloo", "--share-gpu", "--steps", "2", "--warmup", "1"]


def test_two_rank_fista_line_and_global_time_to_tol():
    out = _bench(SHARED + ["--workload", "fista", "--no-cpu-baseline"])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["scaling"] == "strong"
    assert out["config"]["rows_per_gpu"] == 2048 and out["config"]["rows_total"] == 4096
    assert out["weak_scaling"]["rows_per_gpu"] == 4096 and out["weak_scaling"]["value"] > 0
    # the all-reduced objective of the sharded 4096-row batch after 100 iterations
    assert abs(out["objective_after_100"] - 63.609337) <= 2e-6 * 63.609337
    # iterations-to-tol under the reference's rule on ALL rows (ista.py:64,93): 263 at any number of ranks
    t = out["time_to_tol"]
    assert t["iterations"] == 263 and "global over all ranks" in t["rule"] and t["rows_total"] == 4096
    assert out["roofline"]["frac"] > 0 and "gloo" in out["backend"]


def test_two_rank_em_line_has_one_message_per_step_or_stage():
    out = _bench(SHARED + ["--workload", "em"])
    assert out["n_gpus"] == 2 and out["config"]["rows_per_gpu"] == 32768
    ar = out["all_reduce_ms"]
    assert ar["bytes"] == 4 * (1024 * 1024 + 1024 * 256 + 12)
    # beyond 8192 rows per rank: the double-buffered loop, ONE message [A | B | tail] per EM step, no other collective
    assert ar["bytes_sent"] == [4 * (1024 * 1280 + 12)] and ar["per_step"] == 1.0 and ar["bytes_per_step"] == ar["bytes"]
    assert out["em_path"].get("overlapped_steps") == 2 and out["em_path"].get("speculative_sweeps") == 2
    assert not out["em_path"].get("replayed_steps")
    # the first EM steps of config 4 on the whole batch: losses[2] of the reference's run (tests/golden/g4_c4_em.npz)
    import numpy as np
    ref = np.load(os.path.join(ROOT, "tests", "golden", "g4_c4_em.npz"))["c_losses_auto"]
    assert abs(out["objective_last_step"] - float(ref[2])) <= 2e-4
    # 4096 .. 8192 rows per rank: the pipelined M-step -- one message per STAGE of block rows of [A | B] (the head: 512
    # rows; then 256 each, the last with the 12-word tail), the same bytes in all
    out = _bench(SHARED + ["--workload", "em", "--rows", "16384"])
    ar = out["all_reduce_ms"]
    assert out["config"]["rows_per_gpu"] == 8192 and out["em_path"].get("pipelined_steps") == 2
    assert ar["bytes_sent"] == [4 * 256 * 1280, 4 * (256 * 1280 + 12), 4 * 512 * 1280] and ar["per_step"] == 3.0
    assert ar["bytes_per_step"] == ar["bytes"]


def test_two_rank_em_line_at_the_shape_of_config_5():
    out = _bench(SHARED + ["--workload", "em", "--shape", "c5", "--rows", "16384"])
    assert out["config"]["rows_per_gpu"] == 8192 and "d=64 k=256" in out["metric"]
    assert out["all_reduce_ms"]["bytes_sent"] == [4 * (256 * 256 + 256 * 64 + 12)]
    assert out["em_path"].get("overlapped_steps") == 2


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_two_rank_line_search_line(dtype):
    out = _bench(SHARED + ["--workload", "c3", "--dtype", dtype])
    assert out["n_gpus"] == 2 and out["config"]["rows_per_gpu"] == 8192 and out["dtype"] == dtype
    assert len(out["trials"]) == 10
    if dtype == "f32":      # every F <= Q decision on sums over both ranks: the reference's trace on the whole batch
        assert out["trials"] == [5, 3, 5, 4, 4, 4, 4, 3, 5, 5]
        assert abs(out["objective"] - 64.142166) <= 1e-5 * 64.142166
    else:
        assert abs(out["objective"] - 64.151779) <= 2e-3 * 64.151779


def test_one_rank_lines_of_configs_3_and_5():
    for dtype in ("bf16", "f32"):
        out = _bench(["--workload", "c3", "--dtype", dtype, "--steps", "3", "--warmup", "1"])
        assert out["n_gpus"] == 1 and out["roofline"]["peak"] == (2500.0 if dtype == "bf16" else 157.3)
        assert out["roofline"]["flop_per_launch"] == (4 * 10 + 2 * sum(out["trials"])) * 16384 * 256 * 1024
        assert 0 < out["roofline"]["frac"] < 1
    out = _bench(["--workload", "em", "--shape", "c5", "--steps", "5", "--warmup", "2"])
    assert out["n_gpus"] == 1 and out["all_reduce_ms"]["per_step"] == 0.0


# ---- the RCCL arms, executed: ONE rank, a real nccl (= RCCL) process group, LASSO_FORCE_COLLECTIVES ------------------
# `--force-dist` makes the N = 1 run take every multi-rank code path (lasso_amd.parallel._sharded): the in-place
# all-reduce of DEVICE tensors, Ranks.max on a device tensor, the sharded E-step with its stop-rule sums in the M-step
# message, the line search's all-reduce callback through a device buffer.  The first 8-GPU run then differs from code
# that has run on hardware in N only.  Same numbers as the plain one-rank run are required.
FORCED = ["--gpus", "1", "--backend", "nccl", "--force-dist", "--steps", "2", "--warmup", "1"]


def test_one_rank_rccl_group_fista_line():
    out = _bench(FORCED + ["--workload", "fista", "--no-cpu-baseline", "--no-shards", "--no-extras"])
    assert out["n_gpus"] == 1 and "ONE-rank process group" in out["backend"] and out["backend"].startswith("nccl")
    assert abs(out["objective_after_100"] - 63.609337) <= 2e-6 * 63.609337        # sums all-reduced on the device
    t = out["time_to_tol"]                                                        # parallel.sharded_encode: chunks +
    assert t["iterations"] == 263 and "global over all ranks" in t["rule"]        # one all-reduce of the sums each


def test_one_rank_rccl_group_em_line():
    # 4096 .. 8192 rows per rank: the two-stream loop with the PIPELINED M-step, one RCCL message per stage
    plain = _bench(["--workload", "em", "--rows", "8192", "--steps", "2", "--warmup", "1"])
    out = _bench(FORCED + ["--workload", "em", "--rows", "8192"])
    ar = out["all_reduce_ms"]
    assert ar["per_step"] == 3.0 and ar["bytes_per_step"] == 4 * (1024 * 1024 + 1024 * 256 + 12)
    assert out["em_path"].get("pipelined_steps") == 2 and not out["em_path"].get("replayed_steps")
    assert abs(out["objective_last_step"] - plain["objective_last_step"]) <= 2e-6 * plain["objective_last_step"]
    assert plain["all_reduce_ms"]["per_step"] == 0.0 and plain["em_path"].get("pipelined_steps") == 2
    assert plain["em_path"].get("deferred_verdicts") == 2 and not out["em_path"].get("deferred_verdicts")
    # beyond: the two-stream loop with the DOUBLE-BUFFERED dictionary, ONE message [A | B | tail] per EM step
    plain = _bench(["--workload", "em", "--steps", "2", "--warmup", "1"])
    out = _bench(FORCED + ["--workload", "em"])
    ar = out["all_reduce_ms"]
    assert ar["per_step"] == 1.0 and ar["bytes_per_step"] == 4 * (1024 * 1024 + 1024 * 256 + 12)
    assert out["em_path"].get("speculative_sweeps") == 2 and not out["em_path"].get("pipelined_steps")
    assert plain["em_path"].get("speculative_sweeps") == 2 and plain["em_path"].get("deferred_verdicts") == 2
    assert abs(out["objective_last_step"] - plain["objective_last_step"]) <= 2e-6 * plain["objective_last_step"]
    # below 4096 rows per rank: the one-stream loop (objective and Gram product are too short to pay for a second stream)
    out = _bench(FORCED + ["--workload", "em", "--rows", "2048"])
    assert out["all_reduce_ms"]["per_step"] == 1.0
    assert out["em_path"].get("overlapped_steps") == 2 and not out["em_path"].get("pipelined_steps") \
        and not out["em_path"].get("speculative_sweeps")


def test_one_rank_rccl_group_line_search_line():
    out = _bench(FORCED + ["--workload", "c3", "--dtype", "f32"])
    assert "every F<=Q decision on all-reduced sums" in out["config"]["parallelism"]
    assert out["trials"] == [5, 3, 5, 4, 4, 4, 4, 3, 5, 5]                        # lasso_fista_solve_sharded + the callback
    assert abs(out["objective"] - 64.142166) <= 1e-5 * 64.142166


def test_conv_lines_one_and_two_ranks():
    """--workload conv: the images sharded over the ranks (no collective); the same codes -- counted by their nonzeros, up
    to the entries at the threshold: with 128 images per rank the overlap-add follows the two-kernel form's order for
    THAT batch size, another one -- from one rank and from two, an HBM roofline block, the kernel lasso_conv_ista_solve dispatches to."""
    one = _bench(["--workload", "conv", "--steps", "2", "--warmup", "1"])
    two = _bench(SHARED + ["--workload", "conv"])
    assert one["roofline"]["bound"] == "hbm" and one["roofline"]["unit"] == "GB/s" and 0 < one["roofline"]["frac"] < 1
    assert "conv_fused_kernel" in one["roofline"]["kernel"]                   # 256 images: a workgroup per image
    assert "conv_fused_kernel" in two["roofline"]["kernel"]                   # 128 per rank: still a workgroup per image
    assert one["config"]["images_total"] == two["config"]["images_total"] == 256 and two["config"]["images_per_gpu"] == 128
    assert one["nonzeros"] > 0 and abs(one["nonzeros"] - two["nonzeros"]) <= 1e-5 * one["nonzeros"]
    rgb = _bench(["--workload", "conv", "--conv-case", "rgb", "--steps", "2", "--warmup", "1"])
    assert rgb["config"]["images_total"] == 64 and rgb["value"] > 0 and rgb["roofline"]["bound"] == "hbm"


# ---- every roofline block names kernels that its own command dispatches (a stale name survived a round) ----
TRACED = [["--workload", "fista", "--no-cpu-baseline", "--no-shards", "--no-extras"],
          ["--workload", "c3", "--dtype", "f32"], ["--workload", "c3", "--dtype", "bf16"],
          ["--workload", "em", "--rows", "8192"], ["--workload", "em", "--shape", "c5", "--rows", "8192"],
          ["--workload", "conv"], ["--workload", "conv", "--conv-case", "rgb"],
          ["--workload", "cd", "--no-cpu-baseline"]]


@pytest.mark.parametrize("args", TRACED, ids=lambda a: "-".join(x.strip("-") for x in a if x != "--workload"))
def test_roofline_kernel_occurs_in_the_dispatch(args, tmp_path):
    """`roofline.kernel` of a bench line against the rocprofv3 kernel trace of the SAME command: every kernel the
    string names (bench.kernel_tokens) must have been dispatched."""
    import csv
    import glob
    import shutil
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        pytest.skip("rocprofv3 not on this box")
    sys.path.insert(0, ROOT)
    import bench
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["TMPDIR"] = str(tmp_path)
    cmd = [prof, "--kernel-trace", "--output-format", "csv", "-d", str(tmp_path / "t"), "-o", "k", "--",
           sys.executable, BENCH] + args + ["--steps", "2", "--warmup", "1"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    traces = glob.glob(str(tmp_path / "t" / "**" / "*kernel_trace.csv"), recursive=True)
    assert traces, "no kernel trace written"
    names = set()
    for t in traces:
        with open(t) as f:
            names |= {row["Kernel_Name"] for row in csv.DictReader(f)}
    tokens = bench.kernel_tokens(out["roofline"]["kernel"])
    assert tokens, out["roofline"]["kernel"]
    flat = [n.replace(" ", "") for n in names]
    for tok in tokens:
        assert any(tok.replace(" ", "") in n for n in flat), (tok, sorted(names)[:40])
