"""TEST INFRASTRUCTURE: stand-in for `bench.py --worker` (DHQR_BENCH_WORKER) so the supervisor of the multi-GPU bench runs
can be exercised without GPUs.  FAKE_WORKER_PLAN = comma-separated behaviour per attempt:
  ok | hang (progress once, then silence on every rank) | hang1 (only rank 1 goes silent; rank 0 keeps waiting for it)
  | crash1 (rank 1 exits 7; the others wait) | mute (every rank exits 0, nobody prints a result line)"""
import json
import os
import sys
import time

att = json.loads(os.environ["DHQR_BENCH_ATTEMPT"])
plan = os.environ["FAKE_WORKER_PLAN"].split(",")
what = plan[min(att["attempt"], len(plan) - 1)]
rank = int(os.environ.get("RANK", "0"))
print(f"[bench progress] fake worker rank {rank} attempt {att['attempt']}: {what}", file=sys.stderr, flush=True)
if what == "ok":
    time.sleep(0.3)
    if rank == 0:
        print(json.dumps({"metric": "fake", "value": 1.0, "attempt": att["attempt"], "attempt_env": att["env"],
                          "attempts_failed": att["failed"], "saw_rank_env": "RANK" in os.environ,
                          "master_port": os.environ.get("MASTER_PORT"), "transport": os.environ.get("DHQR_TRANSPORT"),
                          "agent_store": os.environ.get("TORCHELASTIC_USE_AGENT_STORE")}), flush=True)
    sys.exit(0)
if what == "mute":
    time.sleep(0.3)
    sys.exit(0)
if what == "crash1" and rank == 1:
    sys.exit(7)
time.sleep(3600)  # hang / hang1 / the peers of a crashed rank: stuck in a collective
