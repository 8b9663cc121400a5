"""-m gpu: `bench.py --gpus N --one-gpu-rehearsal` -- the N-GPU run of the benchmark END TO END on one GPU.

Every rank sits on GPU 0, the torch side talks gloo, the library's RCCL symbols come from tests/fake_rccl.  Nothing in the line is a
measurement; what is rehearsed is everything an 8-GPU lease would otherwise execute for the first time: launch_ranks (fresh child
ranks, one relayed line), the library's communicator with N ranks (`rccl_ranks`), run_sharded -- the north-star configuration FIRST
(ncclAllGather + ncclAllReduce), its guard, the cheaper exchanges validated against its product, the votes, the budget logic, the
fall-backs --, both CG schedules, the parts of an iteration, and what happens when a rank dies in the middle of a timed solve."""
import json
import os
import subprocess
import sys
import time

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def run_bench(*args, env=None, timeout=560):
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=timeout,
                       env=dict(os.environ, **(env or {})))
    lines = [ln for ln in p.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:] + p.stderr[-4000:]
    return p.returncode, json.loads(lines[0]), time.time() - t0, p.stderr


@pytest.mark.parametrize("world,pattern", [(4, "constant_diagonals"), (2, "scrambled")])
def test_rehearsal_end_to_end(world, pattern):
    rc, out, _, err = run_bench("--gpus", str(world), "--one-gpu-rehearsal", "--rows", "2000000", "--band", "65536", "--pattern", pattern,
                                "--steps", "20", "--warmup", "3", "--reps", "3", "--budget-seconds", "400")
    assert rc == 0 and "error" not in out, (out.get("error"), err[-3000:])
    assert out["n_gpus"] == world and out["rccl_ranks"] == world and out["torch_world_size"] == world
    assert "librccl_fake" in out["rccl_library"] and "NO number" in out["rehearsal"]
    assert out["value_rccl_allgather"] > 0 and out["value"] >= out["value_rccl_allgather"] * 0.999 and out["scaling"] == "strong"
    probe = out["comm_probe"]
    tried = probe["configurations_it_per_s"]
    assert "all-gather + rccl all-reduce" in tried
    # the mailboxes connect between processes of one GPU: the direct configurations were validated against the all-gather product too
    assert probe["direct_paths"] == "connected and self-tested", probe["direct_paths"]
    assert any(k.startswith("direct peer writes") for k in tried) and any(k.startswith("neighbour ranges") for k in tried), tried
    for sched in (out["value_rccl_allgather_by_cg_schedule"], out["value_by_cg_schedule"]):
        assert set(sched) == {"classic", "one_reduction"} and all(v > 0 for v in sched.values()), sched
    parts = probe["iteration_parts"]
    for key in ("iteration_us", "ax_in_loop_us", "rest_in_loop_us", "x_exchange_alone_all_gather_us", "x_exchange_alone_neighbour_ranges_us",
                "local_column_product_alone_us", "remote_column_part_alone_us", "serial_sum_us", "overlap_model_us"):
        assert parts[key] > 0, (key, parts)
    per = parts["per_iteration"]
    # one-reduction CG: one product and one reduction over ranks per iteration (+ the set-up's three reductions over 20 iterations)
    assert per["products"] == pytest.approx(1.0, abs=0.11) and 1.0 <= per["rank_reductions"] <= 1.3, per
    assert parts["iteration_us"] == pytest.approx(1e6 / out["value_rccl_allgather"], rel=0.5)       # the north-star configuration is the one taken apart
    assert parts["x_exchange_all_gather_doubles_received"] == (2000000 // world) * (world - 1)
    if pattern == "constant_diagonals":
        assert 0 < parts["x_exchange_neighbour_ranges_doubles_received"] <= 2 * 65536
    assert probe["allreduce_4_doubles_us"] > 0 and probe["ax_with_exchange_us"] > 0
    chk = out["solution_check"]
    assert chk["rel_err_vs_x_true_after_100_iterations"] < 1e-3
    assert "row-block x%d" % world in out["config"]["partition"]


def test_a_rank_that_dies_in_a_timed_solve_leaves_an_error_line_not_a_hang():
    rc, out, seconds, err = run_bench("--gpus", "3", "--one-gpu-rehearsal", "--rows", "1500000", "--band", "65536", "--steps", "400", "--warmup", "3",
                                      "--reps", "2", "--budget-seconds", "300", env={"LCG_BENCH_TEST_DIE_RANK": "1", "FAKE_RCCL_TIMEOUT_S": "5"}, timeout=400)
    assert rc != 0 and out["value"] == 0.0 and out["error"], out
    assert seconds < 200, seconds
