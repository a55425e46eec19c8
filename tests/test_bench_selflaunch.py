"""`python bench.py --gpus 2` end to end on the CPU tier: the self-launcher (one process per rank, RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_* set by bench.py itself, like the reference's one command that drives all its GPUs, /root/reference/ssn_train.py:67),
the gloo rendezvous, the step loop with the product's deferred gradient reduction ('separate' collectives) and with the
bucketed one ('overlapped'), the fences, the max-over-ranks timing and rank 0's single JSON line.

SSN_BENCH_EMULATOR=1 (test tooling, see bench.py) runs the kernels of the heads / STPP / losses / optimizer through the host
emulator and replaces the backbone by a stand-in (the emulated backbone takes minutes per frame; the real one is covered with two
ranks by tests/test_parallel_model.py).  What this test keeps from rotting is the launcher and the distributed control flow."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("collectives", ["separate", "overlapped"])
def test_bench_self_launch_two_ranks(emu_library, collectives):
    env = dict(os.environ, SSN_BENCH_EMULATOR="1", SSN_BENCH_ONE_DEVICE="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--videos-per-gpu", "1",
           "--frame-size", "16", "--collectives", collectives]
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, text=True)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout          # rank 0 prints ONE JSON line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1
    assert out["value"] > 0 and out["unit"] == "proposals/s" and out["scaling"] == "weak"
    assert out["config"]["ranks"] == 2 and out["config"]["collectives"] == collectives and out["config"]["backend"] == "gloo"
    assert out["config"]["global_batch_proposals"] == 16
    assert "EMULATOR" in out["data"]            # nobody can mistake this line for a measurement


@pytest.mark.parametrize("collectives", ["separate", "overlapped"])
def test_bench_self_launch_eight_ranks_with_a_fault_on_rank_5(emu_library, collectives):
    """the `--gpus 8` path -- self-launch, rendezvous, step loop, collectives, fixed-lag fault poll -- with EIGHT processes
    (gloo, host emulator, stand-in backbone).  A range fault injected on rank 5 in step 2 must be seen by ALL ranks behind the same step
    (the word is MAX-reduced inside the step), all eight redo the same three steps, and the run ends with as many updates as steps."""
    env = dict(os.environ, SSN_BENCH_EMULATOR="1", SSN_BENCH_ONE_DEVICE="1", SSN_BENCH_INJECT_FAULT="5:2", OMP_NUM_THREADS="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "5", "--warmup", "1", "--videos-per-gpu", "1",
           "--frame-size", "16", "--collectives", collectives]
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, text=True)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["config"]["ranks"] == 8 and out["config"]["global_batch_proposals"] == 64
    seen = out["distributed"]["ranks_seen"]
    assert sorted(r["rank"] for r in seen) == list(range(8))
    assert out["distributed"]["distinct_devices"] == 8            # (emulator: one host process per rank)
    # every rank saw exactly one fault, behind the SAME step, and redid the same number of steps
    assert [r["guard_faults"] for r in seen] == [1] * 8, seen
    assert len({tuple(r["guard_fault_steps"]) for r in seen}) == 1 and seen[0]["guard_fault_steps"] == [2], seen
    assert [r["redone_steps"] for r in seen] == [3] * 8
    sg = out["scale_guard"]
    assert sg["scale_overflows"] == 1 and sg["steps_with_skipped_update"] == 3 and sg["steps_redone_eagerly"] == 3
    assert sg["effective_updates"] == out["steps"] + out["warmup"]
