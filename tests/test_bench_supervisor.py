"""bench.py --gpus N (N > 1) runs under a supervisor: a worker that stops making progress (a collective that never
completes) or fails is ended on every rank and the run is repeated with a more conservative communication set-up, so the
driver's multi-GPU bench produces a line instead of a timeout.  Exercised here with a stand-in worker, under the same
launcher the driver uses (python -m torch.distributed.run) and in the single-process form."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "helpers", "fake_bench_worker.py")


def _run(plan, nproc, torchrun=True, timeout=180):
    env = dict(os.environ, DHQR_BENCH_WORKER=WORKER, FAKE_WORKER_PLAN=plan, DHQR_BENCH_STALL_S="2", DHQR_BENCH_START_S="30",
               DHQR_BENCH_ATTEMPT_S="60")
    for v in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(v, None)
    port = 29900 + os.getpid() % 90
    cmd = ([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
            "--master-port", str(port)] if torchrun else [sys.executable])
    cmd += [os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), "--steps", "1", "--warmup", "0"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, lines, port


def test_first_attempt_succeeds_one_line():
    r, lines, port = _run("ok", 2)
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["attempt"] == 0 and out["attempts_failed"] == [] and out["saw_rank_env"]
    # the workers rendezvous on their own port with their own store (torchrun's agent store stays the supervisors')
    assert out["master_port"] == str(port + 1) and out["agent_store"] == "False"


def test_hung_collective_falls_back_to_conservative_rccl_setup():
    r, lines, _ = _run("hang1,ok", 2)
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["attempt"] == 1 and out["attempt_env"] == {"DHQR_LANE_CHANNEL": "0", "DHQR_BCAST": "ring"}
    assert len(out["attempts_failed"]) == 1 and "no progress" in out["attempts_failed"][0]["why"]


def test_crash_then_hang_ends_on_the_single_process_peer_copy_run():
    r, lines, _ = _run("crash1,hang,ok", 2)
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["attempt"] == 2 and out["transport"] == "local" and not out["saw_rank_env"]  # one process drives every GPU
    why = [f["why"] for f in out["attempts_failed"]]
    assert "exit code 7" in why[0] and "no progress" in why[1]


def test_every_attempt_failing_is_an_error_not_a_hang():
    r, lines, _ = _run("hang", 2)
    assert r.returncode != 0 and not lines
    assert "every attempt failed" in r.stderr


def test_single_process_form_is_supervised_too():
    r, lines, _ = _run("hang,ok", 2, torchrun=False)
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1 and json.loads(lines[0])["attempt"] == 1


def test_worker_without_a_result_line_is_a_failed_attempt_on_every_rank():
    """every rank exits 0 but rank 0 finds no JSON line: the attempt fails for ALL ranks (rank 0 checks before it publishes its
    verdict), so the ranks enter the next attempt together instead of rank 0 alone waiting at a rendezvous"""
    r, lines, _ = _run("mute,ok", 2, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["attempt"] == 1 and "no result line" in out["attempts_failed"][0]["why"]
