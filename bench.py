#!/usr/bin/env python
"""bench.py -- headline benchmark of the MI355X Householder QR hot path (driver contract).

  python bench.py --gpus N --steps K --warmup W          one process drives the N GPUs (a host thread per GPU)
  python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...     one process per GPU (same C drivers)

A "step" is one complete device-resident QR factorisation of the workload matrix (synthetic
U[0,1) input regenerated on the device before every step; the ~3 ms fill is inside the timed
region, inputs never cross PCIe).  Default workload = BASELINE.json configs[2]: 32768 x 32768
Float64, blocked, panel width 128 (8 GiB, fits one GPU).  For N > 1 the SAME matrix is
block-cyclically column-split over the N ranks ("scaling": "strong", BASELINE configs[3]).
--config unblocked selects configs[1] (8192 x 8192, rank-1 path, HBM roofline).

value = F(m,n) / t with F = 2 m n^2 - 2/3 n^3 (the reference's flop count).
The JSON line also carries ||A-QR||_F/||A||_F of the last factorisation, `roofline` for the
dominant kernel group (hipEvent-timed on the launch stream inside the timed region) and, at
N = 1, `cpu_baseline` = the oracle's restatement of the reference algorithm on the host cores.

N > 1 runs under a SUPERVISOR (this file, no GPU work of its own): every rank's bench.py starts the real run as a child
process and watches its progress lines.  A child that makes no progress for DHQR_BENCH_STALL_S seconds (a collective that
never completes) or fails is killed on every rank and the run is repeated with a more conservative communication set-up:
attempt 0 library defaults -> attempt 1 DHQR_LANE_CHANNEL=0 + DHQR_BCAST=ring (one RCCL communicator, plain ncclBroadcast) ->
attempt 2 one process drives all GPUs with peer copies instead of RCCL (DHQR_TRANSPORT=local).  The JSON line says which
attempt produced it (`attempt`, `attempt_env`, `attempts_failed`).  Nothing is measured by the supervisor.
"""
from __future__ import annotations

import argparse
import glob
import hashlib
import json
import os
import signal
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# the host driver only supports dmabuf IPC; must be in the environment before the HSA runtime starts
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# kernel arguments written straight into device memory: a few microseconds less per launch; the blocked driver issues ~40
# launches per panel (profiles/r05_ab_dev_kernarg.txt: 2048^2 -10 %, 8192^2 -2.6 %, 32768^2 within noise).  Read when the HIP
# runtime initialises, i.e. before torch touches the device; a caller's own setting wins.
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

PEAK_FP64_MFMA_TFLOPS = 78.6   # AMD MI355X datasheet FP64 matrix (== vector) peak; the local
                               # MI355X_MICROARCH.md guide lists no FP64 number
PEAK_HBM_GBPS = 8000.0         # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)

_T0 = time.time()


def progress(label):
    """one line per phase on stderr: what the supervisor of a multi-GPU run watches (a run that stops printing is hung)"""
    print(f"[bench progress] +{time.time() - _T0:.1f}s rank {os.environ.get('RANK', '0')}: {label}", file=sys.stderr, flush=True)


class GpuTelemetry(threading.Thread):
    """Power and shader clock of ONE GPU sampled from the amdgpu hwmon files while the timed region runs (this process's
    own measurement; no literal numbers).  Absent files -> {"error": ...}."""

    def __init__(self, index=0, period_s=0.05, pci=None):
        """pci: "dddd:bb:dd.f" of the device under test (the box may expose more cards than this process may use: the
        hwmon files of another, idle card read 95 MHz); without a match the index-th amdgpu card is taken"""
        super().__init__(daemon=True)
        self.period, self.stop_flag, self.w, self.mhz, self.src = period_s, threading.Event(), [], [], None
        cards, by_pci = [], {}
        for d in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
            try:
                if open(os.path.join(d, "vendor")).read().strip() != "0x1002":
                    continue
            except OSError:
                continue
            hw = sorted(glob.glob(os.path.join(d, "hwmon", "hwmon*")))
            if hw:
                cards.append(hw[0])
                by_pci[os.path.basename(os.path.realpath(d)).lower()] = hw[0]
        self.matched_pci = bool(pci and pci.lower() in by_pci)
        if self.matched_pci or index < len(cards):
            hw = by_pci[pci.lower()] if self.matched_pci else cards[index]
            self.fp = next((f for f in (os.path.join(hw, "power1_average"), os.path.join(hw, "power1_input")) if os.path.exists(f)), None)
            self.ff = os.path.join(hw, "freq1_input") if os.path.exists(os.path.join(hw, "freq1_input")) else None
            self.src = hw
        else:
            self.fp = self.ff = None

    def run(self):
        while not self.stop_flag.is_set():
            try:
                if self.fp:
                    self.w.append(int(open(self.fp).read()) / 1e6)
                if self.ff:
                    self.mhz.append(int(open(self.ff).read()) / 1e6)
            except (OSError, ValueError):
                pass
            self.stop_flag.wait(self.period)

    def result(self):
        self.stop_flag.set()
        if self.is_alive():
            self.join(timeout=2)
        if not self.w and not self.mhz:
            return {"error": "no amdgpu hwmon power1_average / power1_input / freq1_input readable under /sys/class/drm"}
        out = {"source": self.src, "device_matched_by_pci_address": self.matched_pci, "samples": max(len(self.w), len(self.mhz)), "period_s": self.period,
               "note": "sampled by this process during the timed region (whole factorisation steps, all kernels)"}
        if self.w:
            out.update(watts_avg=sum(self.w) / len(self.w), watts_max=max(self.w))
        if self.mhz:
            out.update(sclk_mhz_avg=sum(self.mhz) / len(self.mhz), sclk_mhz_min=min(self.mhz))
        return out


def git_blob_hash(path):
    """git's blob id of a file (sha1 of "blob <len>\0" + content): ties committed counter data to the kernel source"""
    try:
        data = open(path, "rb").read()
    except OSError:
        return None
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


# ---------------------------------------------------------------------------------------------------------------------
# Supervisor of the multi-GPU runs (see the module docstring).  No GPU work, no torch.distributed process group of its
# own: the ranks agree on the fate of an attempt through a key-value store on MASTER_PORT (torchrun's rendezvous port
# serves the launcher's own store on the agent, the per-rank MASTER_PORT is free for the job).
ATTEMPTS = [
    ("library defaults", {}),
    ("one RCCL communicator, plain ncclBroadcast", {"DHQR_LANE_CHANNEL": "0", "DHQR_BCAST": "ring"}),
    ("one process drives every GPU, peer copies instead of RCCL", {"DHQR_TRANSPORT": "local"}),
]


def _kill_tree(proc):
    """end exactly the child we started (it is its own process-group leader) and whatever it spawned"""
    try:
        os.killpg(proc.pid, signal.SIGKILL)
    except (ProcessLookupError, PermissionError):
        pass
    try:
        proc.wait(timeout=30)
    except Exception:
        pass


def supervise(argv):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    spmd = world > 1
    stall_s = float(os.environ.get("DHQR_BENCH_STALL_S", "120"))
    start_s = float(os.environ.get("DHQR_BENCH_START_S", "420"))   # first import of torch on a fresh box: minutes
    total_s = float(os.environ.get("DHQR_BENCH_ATTEMPT_S", "700"))
    worker = os.environ.get("DHQR_BENCH_WORKER")  # tests: a stand-in worker script
    base_port = int(os.environ.get("MASTER_PORT", "29500"))
    store = None
    if spmd:
        from datetime import timedelta
        from torch.distributed import TCPStore
        # torchrun's static rendezvous serves the agent's store on MASTER_PORT and tells the workers to join it as clients
        # (TORCHELASTIC_USE_AGENT_STORE); launched any other way, rank 0 hosts the store
        agent_store = os.environ.get("TORCHELASTIC_USE_AGENT_STORE") == "True"
        store = TCPStore(os.environ.get("MASTER_ADDR", "127.0.0.1"), base_port, None if agent_store else world,
                         rank == 0 and not agent_store, timeout=timedelta(seconds=start_s))
    failed = []
    for k, (what, extra) in enumerate(ATTEMPTS):
        single_process = "DHQR_TRANSPORT" in extra
        env = dict(os.environ, **extra)
        env["DHQR_BENCH_ATTEMPT"] = json.dumps({"attempt": k, "what": what, "env": extra, "failed": failed})
        proc = None
        if spmd and single_process:
            # last resort: rank 0 alone starts ONE worker that drives all GPUs with its rank threads; the other ranks idle
            for v in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "TORCHELASTIC_RUN_ID"):
                env.pop(v, None)
        elif spmd:
            env["MASTER_PORT"] = str(base_port + 1 + k)  # the children's own rendezvous: a fresh store per attempt,
            env["TORCHELASTIC_USE_AGENT_STORE"] = "False"  # hosted by the child of rank 0
        if not (spmd and single_process and rank != 0):
            cmd = ([sys.executable, worker] if worker else [sys.executable, os.path.abspath(__file__), "--worker"]) + argv
            proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
        state = {"last": time.time(), "seen": False, "out": []}

        def pump(stream, is_out):
            for line in stream:
                state["last"] = time.time()
                if is_out:
                    state["out"].append(line)
                else:
                    if line.startswith("[bench progress]"):
                        state["seen"] = True
                    sys.stderr.write(line)
                    sys.stderr.flush()

        threads = []
        if proc is not None:
            threads = [threading.Thread(target=pump, args=(proc.stdout, True), daemon=True),
                       threading.Thread(target=pump, args=(proc.stderr, False), daemon=True)]
            [t.start() for t in threads]
        t_begin = time.time()
        verdict = "ok" if proc is None else None
        abort_key = f"dhqr_bench/a{k}_abort"
        while verdict is None:
            rc = proc.poll()
            now = time.time()
            if rc is not None:
                verdict = "ok" if rc == 0 else f"exit code {rc}"
            elif now - state["last"] > (stall_s if state["seen"] else start_s):
                verdict = f"no progress for {now - state['last']:.0f} s"
            elif now - t_begin > total_s:
                verdict = f"attempt longer than {total_s:.0f} s"
            elif store is not None and store.check([abort_key]):
                verdict = "a peer rank gave up"
            else:
                time.sleep(0.25)
        if proc is not None and verdict != "ok":
            print(f"[bench supervisor] rank {rank}: attempt {k} ({what}): {verdict}; ending the worker", file=sys.stderr, flush=True)
            _kill_tree(proc)
        [t.join(timeout=5) for t in threads]
        # rank 0 owns the result line: a worker that exits 0 without printing one is a FAILED attempt for every rank (checked
        # before the verdict is published, so the ranks cannot diverge on whether the attempt counts)
        result_line = None
        if verdict == "ok" and rank == 0:
            lines = [ln for ln in state["out"] if ln.startswith("{")]
            if lines:
                result_line = lines[-1]
            else:
                verdict = "the worker printed no result line"
        all_ok = verdict == "ok"
        if store is not None:
            if not all_ok:
                store.set(abort_key, "1")
            store.set(f"dhqr_bench/a{k}_r{rank}", verdict)
            try:
                verdicts = [store.get(f"dhqr_bench/a{k}_r{r}").decode() for r in range(world)]  # blocks until every rank has reported
            except Exception as e:
                verdicts = [f"rank verdicts unavailable: {e!r}"]
            all_ok = all(v == "ok" for v in verdicts)
            if not all_ok:
                verdict = "; ".join(f"rank {r}: {v}" for r, v in enumerate(verdicts) if v != "ok")
        if all_ok:
            if rank == 0:
                sys.stdout.write(result_line)
                sys.stdout.flush()
            return 0
        failed.append({"attempt": k, "what": what, "env": extra, "why": verdict})
        if rank == 0:
            print(f"[bench supervisor] attempt {k} ({what}) failed: {verdict}", file=sys.stderr, flush=True)
    if rank == 0:
        print(json.dumps({"error": "every attempt failed", "attempts_failed": failed}), file=sys.stderr, flush=True)
    return 3


def flops_qr(m, n):
    return 2.0 * m * n * n - 2.0 / 3.0 * n ** 3


def cpu_baseline(m, n, budget_s=20.0):
    """Oracle (= line-by-line port of the reference's unblocked algorithm, OpenMP over trailing
    columns like the reference's @batch) on a bounded sample: the first `cols` reflectors of the
    same m x n workload, each applied to all trailing columns."""
    import numpy as np
    import psutil
    from oracle import dhqr_oracle as orc
    note = ""
    need = m * n * 8
    avail = psutil.virtual_memory().available
    if need * 1.2 > avail:
        scale = 2
        while (m // scale) * (n // scale) * 8 * 1.2 > avail:
            scale *= 2
        m, n = m // scale, n // scale
        note = f" (host RAM too small for the full matrix: shape reduced to {m}x{n})"
    H = np.empty((m, n), order="F")
    orc.lib().dhqr_oracle_fill(orc._ptr(H), m, n, m, 0)
    alpha = np.zeros(n)
    done, t_total, fl = 0, 0.0, 0.0
    chunk = 4
    while t_total < budget_s and done < n:
        c = min(chunk, n - done)
        # prefix(j0..j0+c): the C entry point always starts at column 0, so run on the sub-block
        sub = H[done:, done:]
        t0 = time.perf_counter()
        orc.lib().dhqr_oracle_householder_prefix(orc._ptr(sub), m - done, n - done, m, orc._ptr(alpha[done:]), c)
        t_total += time.perf_counter() - t0
        fl += sum(4.0 * (m - j) * (n - j - 1) for j in range(done, done + c))
        done += c
        if t_total > 0 and t_total / done * chunk * 2 < budget_s / 4:
            chunk *= 2
    return {
        "value": fl / t_total / 1e9, "unit": "GFLOP/s", "cores": orc.num_threads(), "kind": "port",
        "sample": f"first {done} of {n} reflectors of the {m}x{n} unblocked factorisation "
                  f"(each applied to every trailing column), {t_total:.1f} s{note}; restatement of "
                  "src/DistributedHouseholderQR.jl:122-148,198-213 (Julia is not installed)",
        "seconds": t_total,
    }


def cpu_baseline_lapack(n=4096, timeout_s=90):
    """'Good CPU' reference point (what test/runtests.jl:53 compares against): LAPACK dgeqrf from SciPy's
    OpenBLAS on the host cores (oracle/lapack_bench.py, in a subprocess: a reported baseline must never be able to take the
    bench line down).  Not the reference's algorithm (blocked, BLAS-3)."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "lapack_bench.py"), str(n)], capture_output=True,
                           text=True, timeout=timeout_s)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
        return json.loads(line)
    except Exception as e:  # a reported baseline, never fatal
        return {"error": repr(e)[:200]}


def cpu_baseline_distributed(procs=2, orders=(512, 2048, 4096), timeout_s=120):
    """The reference's Distributed.jl STRUCTURE on the host cores (BASELINE configs[0]: 512 x 512, nprocs = 2, and larger
    orders): oracle/dist_oracle.py run as `procs` gloo processes x (cores / procs) OpenMP threads -- contiguous column
    blocks, one broadcast of the dense column per reflector (src:115-148).  Julia itself is not installed."""
    import subprocess
    cores = os.cpu_count() or 2
    env = dict(os.environ, OMP_NUM_THREADS=str(max(1, min(64, cores // procs))))
    port = 29700 + os.getpid() % 200
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={procs}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "oracle", "dist_bench.py")] + [str(n) for n in orders]
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout_s)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("[{")][-1]
        return json.loads(line)
    except Exception as e:  # a reported baseline, never fatal
        return {"error": repr(e)[:200]}


def attempt_fields(attempt):
    if not attempt:
        return {}
    return {"attempt": attempt["attempt"], "attempt_what": attempt["what"], "attempt_env": attempt["env"], "attempts_failed": attempt["failed"]}


def tallskinny(args, pkg, torch, dist, world, rank, local_rank, dev, spmd, attempt=None):
    """BASELINE configs[4]: 262144 x 4096 Float64, rows split over the ranks (dhqr_rs_* / dhqr_mg_rs_*)."""
    m = args.m or 262144
    n = args.n or 4096
    mg = q = None
    if spmd:
        q = pkg.RowSplitQR(m, n, comm=pkg.Communicator.from_torch(pkg.get_context(local_rank)))
        progress("RCCL communicators up (row split)")

        def step():
            q.fill(0)
            q.factor()
    else:
        devices = list(range(world)) if (world > 1 or not args.logical_ranks) else [0] * args.logical_ranks
        mg = pkg.MultiGpuQR(devices=devices)
        progress(f"multi-GPU handle up, transport {mg.transport}")
        mg.rs_alloc(m, n)

        def step():
            mg.rs_fill(0)
            mg.rs_factor()

    def barrier():
        for d in range(torch.cuda.device_count() if mg is not None else 1):
            torch.cuda.synchronize(d if mg is not None else local_rank)
        if spmd:
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(args.warmup):
        step()
        barrier()
        progress(f"warm-up step {i + 1}/{args.warmup}")
    barrier()
    if mg is not None:
        mg.reset_stats()
        mg.set_profiling(True)
    cc0 = mg.comm_counters(0) if mg is not None else q.comm.counters()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    progress(f"{args.steps} timed steps done")
    cc1 = mg.comm_counters(0) if mg is not None else q.comm.counters()
    st = None
    if mg is not None:
        st = mg.stats(0)
        mg.set_profiling(False)
    if spmd:
        t = torch.tensor([dt], dtype=torch.float64)  # host tensor: the control plane is gloo
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    resid = None
    if not args.no_residual:
        resid = mg.rs_residual(0) if mg is not None else q.residual(0)
    value = flops_qr(m, n) / (dt / args.steps) / 1e9
    nranks = world if world > 1 else (args.logical_ranks or 1)
    out = {
        "metric": "QR GFLOP/s (F = 2mn^2 - 2/3 n^3), ||A-QR||/||A|| alongside",
        "value": value, "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{m}x{n} Float64 tall-skinny QR, row split (BASELINE configs[4])", "m": m, "n": n,
                   "nb": 128, "parallelism": f"rows split x{nranks} (128-row aligned slabs), all-reduce of Gram matrices and V'C "
                                             f"partial dots" + (f", transport {mg.transport}" if mg is not None else ", RCCL")},
        "residual": resid,
        # count and volume of the all-reduces of cross-partition partial dots, per factorisation and
        # rank, as the driver issued them (at one rank they move nothing; the count and the sizes do not depend on N)
        "allreduce_per_step": {"count": (cc1["n_allreduce"] - cc0["n_allreduce"]) / args.steps,
                               "bytes": (cc1["bytes_allreduce"] - cc0["bytes_allreduce"]) / args.steps},
        **attempt_fields(attempt),
        **({"rccl_nranks": (q.comm.rccl_nranks() if spmd else mg.rccl_nranks())} if nranks > 1 else {}),
        "roofline": {"bound": "mfma", "achieved": value / 1e3 / max(world, 1), "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s",
                     "frac": value / 1e3 / max(world, 1) / PEAK_FP64_MFMA_TFLOPS, "traffic": None,
                     "kernel": "whole row-split factorisation per GPU (not a single kernel)"},
    }
    if st is not None:
        out["phase_ms_per_step"] = {k: st[k] / args.steps for k in st if k.startswith("ms_") and st[k] > 0}
        out["panels_fast_fallback"] = [st["panels_fast"], st["panels_fallback"]]
    if rank == 0:
        print(json.dumps(out), flush=True)
    if mg is not None:
        mg.close()
    if spmd:
        dist.barrier()
        dist.destroy_process_group()


def emit(out, rank=0):
    if rank == 0:
        print(json.dumps(out), flush=True)


def pmc_traffic(symbol, m, n, nb, launches, work):
    """(HBM traffic of one factorisation step, reason-if-None) of a roofline kernel from the committed counter run
    profiles/pmc_traffic_current.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over tools/pmc_driver,
    gfx950 x2 FETCH correction, calibrated on a streaming kernel).  Counters cannot be read from inside this process, so the
    figure is the one measured for the SAME kernel symbol on the SAME workload -- and only while the kernel's source files
    still have the git blob ids the counter run was stamped with."""
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic_current.json")))
    except Exception as e:
        return None, f"profiles/pmc_traffic_current.json unreadable: {e!r}"
    workload = (f"blocked {m}x{n} nb=128" if nb else f"unblocked {m}x{n}")
    for e in pm.get("entries", []):
        if e["kernel_symbol"] != symbol or not e["workload"].startswith(workload):
            continue
        for src in e["sources"]:
            have, want = git_blob_hash(os.path.join(ROOT, src)), pm.get("source_hashes", {}).get(src)
            if have != want:
                return None, f"committed counters were taken with {src} at blob {str(want)[:12]}, this tree has {str(have)[:12]}: re-run the counter passes of tools/gpu_r5_evidence.sh and tools/pmc_stamp.py"
        # bytes of ONE factorisation step as the counters measured them: read + written by the launches the counter passes
        # cover (`counted`: the wide launches of the symbol; the narrow look-ahead launches of the same template are left out)
        info = {"bytes_per_step": (e["read_GB"] + e["write_GB"]) * 1e9, "counted_launches_per_step": e["launches"],
                "algorithmic_bytes_per_step_of_those": (e["algorithmic_GB"] * 1e9 if e.get("algorithmic_GB") is not None else None),
                "ratio_to_algorithmic": e.get("ratio_to_algorithmic")}
        return info, None
    return None, f"no counter run committed for {symbol} on {workload}"


def roofline_groups(st, steps, m=0, n=0, nb=0):
    """per-kernel-group roofline entries from the hipEvent statistics of ONE rank"""
    groups = []
    if st["ms_gemm_avw"] > 0:
        # one timed group = ONE wide subtraction launch on the caller's stream: k_gemm_nn_quad (four panels, K = 512) while
        # quad steps run, k_gemm_nn_sub (K = 256 / 128) for the pairs / single panels of the tail (rocprofv3 lists the
        # narrow look-ahead launches of the same templates on the second stream as well: profiles/*_by_stream.csv)
        groups.append(dict(kernel="k_gemm_nn_quad / k_gemm_nn_sub (A -= V*W, K = 512 / 256, FP64 MFMA)", symbol="k_gemm_nn_quad", bound="mfma", ms=st["ms_gemm_avw"],
                           launches=st["n_gemm_avw"], work=st["flops_gemm_avw"]))
        # one timed group = the TN launches of one wide update (one k_gemm_tn2 per pair of panels) + their split-K reductions
        groups.append(dict(kernel="k_gemm_tn2 / k_gemm_tn (W = [V_a V_b]'*A, FP64 MFMA)", symbol="k_gemm_tn2", bound="mfma", ms=st["ms_gemm_vta"],
                           launches=st["n_gemm_vta"], work=st["flops_gemm_vta"]))
    if st["ms_panel"] > 0 and st["bytes_panel"] > 0:  # (drivers that do not count the panel's algorithmic bytes get no entry: no placeholder)
        # a LATENCY chain (two single-workgroup kernels per panel), not a bandwidth kernel: `ms` is the elapsed time of the panel
        # factorisations on the lane INCLUDING their wait for CUs behind the wide launches (DESIGN.md section 3 "The chain
        # budget": ~0.29 ms per panel on an idle chip), priced against the reference's in-panel HBM traffic (16 B per element
        # touched per reflector); the narrow updates / cross terms of the lane are not timed (two event records per section on the critical chain)
        groups.append(dict(kernel="panel lane: the panel factorisations (Gram / Cholesky + replay / reconstruction kernels, dhqr_recon.h); elapsed incl. waits for CUs",
                           symbol="panel lane", bound="hbm", ms=st["ms_panel"], launches=st["n_panel"], work=st["bytes_panel"]))
    if st["ms_rank1"] > 0:
        # work = algorithmic HBM bytes of the launches AS IMPLEMENTED: a pass applies K reflectors to every trailing column
        # it loads and stores once (16 B per element and pass = 16/K B per element and reflector)
        groups.append(dict(kernel="k_rankk_fused (reflector apply, K reflectors per pass over the trailing columns)",
                           symbol="k_rankk_fused", bound="hbm", ms=st["ms_rank1"], launches=st["n_rank1"], work=st["bytes_rank1"]))
    rl_all = []
    for gr in groups:
        if gr["bound"] == "mfma":
            ach, peak, unit = gr["work"] / gr["ms"] / 1e9, PEAK_FP64_MFMA_TFLOPS, "TFLOP/s"
        else:
            ach, peak, unit = gr["work"] / gr["ms"] / 1e6, PEAK_HBM_GBPS, "GB/s"
        tinfo, why = pmc_traffic(gr["symbol"], m, n, nb, gr["launches"], gr["work"])
        # ONE definition of "launch" for every per-launch field: the hipEvent groups this run timed (`launches` of them in
        # `steps` steps, `avg_launch_ms` each).  traffic = counter bytes of a step / timed groups of a step, so that
        # traffic / avg_launch_ms = traffic_bytes_per_step / (total_ms / steps) = the HBM rate the group sustained.
        per_step = gr["launches"] / max(1, steps)
        entry = {"kernel": gr["kernel"], "bound": gr["bound"], "achieved": ach, "peak": peak, "unit": unit,
                 "frac": ach / peak, "traffic": (tinfo["bytes_per_step"] / per_step if tinfo else None),
                 **({"traffic_null_because": why} if tinfo is None else {}),
                 "launches": gr["launches"], "launches_per_step": per_step,
                 "avg_launch_ms": gr["ms"] / max(1, gr["launches"]), "total_ms": gr["ms"], "ms_per_step": gr["ms"] / max(1, steps)}
        if tinfo:
            ratio = tinfo["ratio_to_algorithmic"]
            if ratio is None and gr["bound"] == "hbm" and gr["work"] > 0:
                # the counter file has no algorithmic figure for this group (the panel lane): this run's own count -- the
                # reference's in-panel traffic, 16 B per element touched per reflector (bytes_panel) -- prices it
                ratio = tinfo["bytes_per_step"] / (gr["work"] / max(1, steps))
            entry.update(traffic_bytes_per_step=tinfo["bytes_per_step"],
                         traffic_counted_launches_per_step=tinfo["counted_launches_per_step"],
                         traffic_ratio_to_algorithmic=ratio,
                         traffic_GBps=tinfo["bytes_per_step"] / (gr["ms"] / max(1, steps)) / 1e6)
        rl_all.append(entry)
    # the north star grades the trailing update: the DOMINANT (largest total time) MFMA group
    mf = [r for r in rl_all if r["bound"] == "mfma"]
    dom = max(mf, key=lambda r: r["total_ms"]) if mf else (max(rl_all, key=lambda r: r["total_ms"]) if rl_all else None)
    return dom, rl_all


def also_unblocked(pkg, torch, ctx, dev, steps=3, warmup=1, n=8192):
    """BASELINE configs[1] beside the headline line: n x n Float64, unblocked (the reference's operation order, K reflectors
    per pass over the trailing columns), device-resident, same timing discipline (refill inside the timed region)"""
    import ctypes
    m = n
    L = pkg._lib.lib()
    A = pkg.empty_colmajor(m, n, dev)
    alpha = torch.zeros(n, dtype=torch.float64, device=dev)

    def step():
        ctx.use_torch_stream()
        pkg._lib.check(L.dhqr_fill_uniform_f64(ctx.handle, ctypes.c_void_p(A.data_ptr()), m, n, m, 0, m, 0, pkg.NB, 1, 0))
        pkg.householder_(A, alpha, nb=0)

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    ctx.reset_stats()
    ctx.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = ctx.stats()
    ctx.set_profiling(False)
    A0 = pkg.rand_colmajor(m, n, 0, dev)
    resid = pkg.residual(pkg.DistributedHouseholderQRStruct(A, alpha), A0)
    dom, _ = roofline_groups(st, steps, m, n, 0)
    ms = dt / steps * 1e3
    return {"config": {"workload": f"{m}x{n} Float64 dense QR, unblocked rank-1 (BASELINE configs[1])", "m": m, "n": n, "nb": 0},
            "value": flops_qr(m, n) / (dt / steps) / 1e9, "unit": "GFLOP/s", "ms_per_step": ms, "steps": steps, "warmup": warmup,
            "residual": resid, "roofline": dom,
            "reflectors_per_pass": 16.0 * sum((m - j) * (n - j - 1) for j in range(n)) * steps / st["bytes_rank1"] if st["bytes_rank1"] > 0 else None}


def cpu_baseline_lapack_ls(timeout_s=150):
    """LAPACK `qr!(A) \\ b` (geqrf + ormqr + trtrs) at the reference's seven shapes on the host cores -- the time
    test/runtests.jl:55-56,87-89 divides by (oracle/lapack_ls_bench.py, in a subprocess; the shapes finished in time)."""
    import subprocess
    try:
        p = subprocess.Popen([sys.executable, os.path.join(ROOT, "oracle", "lapack_ls_bench.py")], stdout=subprocess.PIPE,
                             stderr=subprocess.DEVNULL, text=True)
        try:
            so, _ = p.communicate(timeout=timeout_s)
        except subprocess.TimeoutExpired:
            p.kill()
            so, _ = p.communicate()
        lines = [ln for ln in (so or "").splitlines() if ln.startswith("{")]
        return json.loads(lines[-1]) if lines else {"error": "no shape finished"}
    except Exception as e:  # a reported baseline, never fatal
        return {"error": repr(e)[:200]}


def also_solve(pkg, torch, dev, lapack=True):
    """SURVEY 8 f1, "device-side solve performance": (1) dhqr_solve_f64 (b <- Q'b + back substitution, csrc/dhqr_qtb.h) on
    factors already in HBM: ms, algorithmic GB/s (one pass over V + one over R) and the fraction of the 8 TB/s roof;
    (2) the expression the reference's test file times, `qr!(A) \\ b` on HOST arrays (test/runtests.jl:59,66), at its seven
    shapes beside LAPACK's geqrf + ormqr + trtrs on the host cores -- the ratio test/runtests.jl:87-89 prints."""
    import numpy as np
    out = {"device_resident": [], "qr_ldiv_host_arrays": []}
    for m, n in ((4400, 4000), (8192, 8192), (32768, 32768)):
        A = pkg.rand_colmajor(m, n, 0, dev)
        H = pkg.qr_(A, nb=128)
        b = pkg.rand_vector_device(m, 1, dev)
        x = pkg.ldiv(H, b)  # warm-up: workspaces
        ts = []
        for _ in range(5):
            bb = b.clone()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            pkg.solve_householder_(bb, H.A, H.α)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ms = min(ts)
        byts = 8.0 * (m * n - n * n / 2.0) + 4.0 * n * n
        A0 = pkg.rand_colmajor(m, n, 0, dev)
        ne = float((A0.T @ (A0 @ x - b)).norm() / (A0.T @ b).norm())
        out["device_resident"].append({"m": m, "n": n, "ms": ms, "algorithmic_bytes": byts, "GBps": byts / ms / 1e6,
                                       "frac_of_8TBps": byts / ms / 1e6 / 8000.0, "normal_eq_rel": ne})
        del A, A0, H, b, x
        torch.cuda.empty_cache()
    lp = cpu_baseline_lapack_ls() if lapack else {}
    lps = {(d["m"], d["n"]): d for d in lp.get("shapes", [])}
    for m, n in ((110, 100), (220, 200), (440, 400), (880, 800), (1100, 1000), (2200, 2000), (4400, 4000)):
        A0 = np.asfortranarray(pkg.rand_colmajor(m, n, 0, dev).cpu().numpy())  # the library's generator (= the LAPACK leg's inputs)
        b0 = pkg.rand_vector_device(m, 1, dev).cpu().numpy()
        best = None
        for _ in range(3):
            A = A0.copy(order="F")
            t0 = time.perf_counter()
            x = pkg.ldiv(pkg.qr_(A), b0)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        r = A0 @ x - b0
        e = {"m": m, "n": n, "seconds": best, "normal_eq": float(np.linalg.norm(A0.T @ r))}
        if (m, n) in lps:
            e["lapack_seconds"] = lps[(m, n)]["seconds"]
            e["times_longer_than_lapack"] = best / lps[(m, n)]["seconds"]
            e["lapack_normal_eq"] = lps[(m, n)]["normal_eq"]
        out["qr_ldiv_host_arrays"].append(e)
    out["lapack_threads"] = lp.get("threads")
    if "error" in lp:
        out["lapack_error"] = lp["error"]
    return {"config": {"workload": "solve: dhqr_solve_f64 on resident factors; qr!(A) \\ b on host arrays at the reference's seven shapes"},
            **out}


def also_sizes(pkg, torch, ctx, dev, sizes=(8192, 16384), steps=8, warmup=3):
    """the blocked path below the headline size, where the panel chain -- not the GEMMs -- bounds the run (DESIGN.md section 3
    "The panel chain"): n x n Float64, nb = 128, device-resident, refill inside the timed region like the headline"""
    import ctypes
    L = pkg._lib.lib()
    out = []
    for n in sizes:
        A = pkg.empty_colmajor(n, n, dev)
        alpha = torch.zeros(n, dtype=torch.float64, device=dev)

        def step():
            ctx.use_torch_stream()
            pkg._lib.check(L.dhqr_fill_uniform_f64(ctx.handle, ctypes.c_void_p(A.data_ptr()), n, n, n, 0, n, 0, pkg.NB, 1, 0))
            pkg.householder_(A, alpha, nb=pkg.NB)

        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        resid = pkg.residual(pkg.DistributedHouseholderQRStruct(A, alpha), pkg.rand_colmajor(n, n, 0, dev))
        out.append({"m": n, "n": n, "nb": pkg.NB, "ms_per_step": dt * 1e3, "value": flops_qr(n, n) / dt / 1e9, "unit": "GFLOP/s",
                    "frac_of_mfma_peak": flops_qr(n, n) / dt / 1e12 / PEAK_FP64_MFMA_TFLOPS, "residual": resid})
        del A, alpha
        torch.cuda.empty_cache()
    return {"config": {"workload": "blocked Float64 QR below the headline size (chain-bound sizes)"}, "sizes": out}


def host_in_out(pkg, torch, dev, m, n, reps=2):
    """the PCIe-inclusive drop-in call `qr!(A::Matrix)` = dhqr_qr_f64 on a HOST matrix (pageable numpy memory, as a Julia
    Matrix would be): copy up, factorisation, copy down (the staged / overlapped form of csrc/dhqr_hostio.h is
    DHQR_HOSTIO=1: slower on this stack).  Reported beside the device-resident time; never the headline `value`."""
    import numpy as np
    Ad = pkg.rand_colmajor(m, n, 0, dev)
    A0 = Ad.cpu().numpy()           # column-major view of the same synthetic input, on the host
    del Ad
    torch.cuda.empty_cache()
    if not A0.flags.f_contiguous:
        A0 = np.asfortranarray(A0)
    ts = []
    H = None
    for _ in range(reps + 1):       # the first call also allocates the pinned staging buffers and the device copy
        del H                       # free the previous result OUTSIDE the timed statement (munmap of 8 GiB: 0.3 s)
        A = A0.copy(order="F")
        t0 = time.perf_counter()
        H = pkg.qr_(A, nb=128)
        ts.append((time.perf_counter() - t0) * 1e3)
    v2 = float((H.A[n // 2:, n // 2] ** 2).sum())
    return {"ms": min(ts[1:]), "first_call_ms": ts[0], "calls": reps, "gflops": flops_qr(m, n) / (min(ts[1:]) / 1e3) / 1e9,
            "check_v_norm2": v2, "note": "wall time of dhqr_qr_f64 on pageable host memory (upload, factorisation, download; DHQR_HOSTIO=1 overlaps the download)"}


def also_tallskinny(pkg, torch, steps=3, warmup=1, m=262144, n=4096):
    """BASELINE configs[4] at world size 1 beside the headline line: the row-split driver (all-reduces issued and counted,
    moving nothing at one rank) on the full 262144 x 4096 shape"""
    mg = pkg.MultiGpuQR(devices=[0])
    try:
        mg.rs_alloc(m, n)
        for _ in range(warmup):
            mg.rs_fill(0)
            mg.rs_factor()
        torch.cuda.synchronize()
        mg.reset_stats()
        mg.set_profiling(True)
        cc0 = mg.comm_counters(0)
        t0 = time.perf_counter()
        for _ in range(steps):
            mg.rs_fill(0)
            mg.rs_factor()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        cc1 = mg.comm_counters(0)
        st = mg.stats(0)
        mg.set_profiling(False)
        resid = mg.rs_residual(0)
    finally:
        mg.close()
    value = flops_qr(m, n) / (dt / steps) / 1e9
    _, rl_all = roofline_groups(st, steps)
    return {"config": {"workload": f"{m}x{n} Float64 tall-skinny QR, row split at world size 1 (BASELINE configs[4])", "m": m, "n": n, "nb": 128},
            "value": value, "unit": "GFLOP/s", "ms_per_step": dt / steps * 1e3, "steps": steps, "warmup": warmup, "residual": resid,
            "allreduce_per_step": {"count": (cc1["n_allreduce"] - cc0["n_allreduce"]) / steps,
                                   "bytes": (cc1["bytes_allreduce"] - cc0["bytes_allreduce"]) / steps},
            "roofline": {"bound": "mfma", "achieved": value / 1e3, "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": value / 1e3 / PEAK_FP64_MFMA_TFLOPS, "traffic": None,
                         "kernel": "whole row-split factorisation (not a single kernel)"},
            "roofline_all": [{k: r[k] for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "launches_per_step", "ms_per_step")} for r in rl_all]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=["blocked", "unblocked", "tallskinny"], default="blocked",
                    help="blocked = BASELINE configs[2]/[3] (default, the metric's configuration); unblocked = configs[1]; "
                         "tallskinny = configs[4] (262144x4096, rows split over the ranks, all-reduce of partial dots)")
    ap.add_argument("--n", type=int, default=0, help="matrix order (default 32768 blocked / 8192 unblocked)")
    ap.add_argument("--m", type=int, default=0, help="rows (default = n)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-residual", action="store_true")
    ap.add_argument("--no-also", action="store_true",
                    help="skip the other single-GPU configurations (unblocked 8192^2, row split 262144x4096 at world size 1) "
                         "that the default 1-GPU run appends to its line as `also`")
    ap.add_argument("--logical-ranks", type=int, default=0,
                    help="development: run R ranks of the multi-GPU driver on ONE GPU (in-process peer-copy transport)")
    ap.add_argument("--worker", action="store_true", help="internal: the measured run itself (started by the supervisor of N > 1)")
    args = ap.parse_args()

    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    if (args.gpus > 1 or env_world > 1) and not args.worker:
        if env_world > 1 and env_world != args.gpus:
            raise SystemExit(f"WORLD_SIZE={env_world} but --gpus {args.gpus}")
        sys.exit(supervise([a for a in sys.argv[1:] if a != "--worker"]))

    progress("start")
    import torch
    import torch.distributed as dist
    import __graft_entry__ as g
    pkg = g.import_package()
    progress("torch and libdhqr loaded")
    attempt = json.loads(os.environ.get("DHQR_BENCH_ATTEMPT", "null"))
    if args.gpus > 1 and not args.logical_ranks:
        ndev = torch.cuda.device_count()
        if ndev < args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus}: only {ndev} GPU(s) visible to this process "
                             f"(HIP_VISIBLE_DEVICES={os.environ.get('HIP_VISIBLE_DEVICES')!r}, ROCR_VISIBLE_DEVICES={os.environ.get('ROCR_VISIBLE_DEVICES')!r})")

    # Launch modes (same C drivers underneath):
    #   python bench.py --gpus N                  ONE process drives N GPUs (dhqr_mg_*: a host thread per device, RCCL
    #                                             communicators from ncclCommInitAll)
    #   torchrun --nproc-per-node N bench.py ...  one process per GPU (dhqr_cs_* over ncclCommInitRank, the unique id
    #                                             shipped through torch.distributed)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    spmd = env_world > 1
    world = args.gpus
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if spmd:
        # control plane only (unique id, barriers, max of the times): gloo, so that the job holds exactly the RCCL
        # communicators libdhqr creates; the data path is RCCL inside the library
        dist.init_process_group(os.environ.get("DHQR_BENCH_PG", "gloo"))
        progress("process group up")

    if args.config == "tallskinny":
        return tallskinny(args, pkg, torch, dist, world, rank, local_rank, dev, spmd, attempt)
    nb = 128 if args.config == "blocked" else 0
    n = args.n or (32768 if nb else 8192)
    m = args.m or n
    seed = 0
    ctx = pkg.get_context(local_rank)
    if nb == 0 and (world > 1 or args.logical_ranks):
        raise SystemExit("the unblocked configuration is single-GPU only")

    mode = "single"
    mg = q = None
    if spmd:
        mode = "spmd"
        q = pkg.ColumnCyclicQR(m, n, comm=pkg.Communicator.from_torch(ctx))
        progress("RCCL communicators up (column split)")

        def step():
            q.fill(seed)
            q.factor()
    elif world > 1 or args.logical_ranks:
        mode = "mg"
        devices = list(range(world)) if world > 1 else [0] * args.logical_ranks
        mg = pkg.MultiGpuQR(devices=devices)
        progress(f"multi-GPU handle up, transport {mg.transport}")
        mg.alloc(m, n)

        def step():
            mg.fill(seed)
            mg.factor()
    else:
        A = pkg.empty_colmajor(m, n, dev)
        alpha = torch.zeros(n, dtype=torch.float64, device=dev)
        L = pkg._lib.lib()
        import ctypes

        def step():
            ctx.use_torch_stream()
            pkg._lib.check(L.dhqr_fill_uniform_f64(ctx.handle, ctypes.c_void_p(A.data_ptr()), m, n, m, seed, m, 0,
                                                   pkg.NB, 1, 0))
            pkg.householder_(A, alpha, nb=nb)

    def barrier():
        for d in range(torch.cuda.device_count() if mode == "mg" else 1):
            torch.cuda.synchronize(d if mode == "mg" else local_rank)
        if spmd:
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(args.warmup):
        step()
        barrier()
        progress(f"warm-up step {i + 1}/{args.warmup}")
    barrier()
    if mode == "mg":
        mg.reset_stats()
        mg.set_profiling(True)
        for r in range(mg.ndev):
            mg.comm_timing(r, on=1)
    else:
        ctx.reset_stats()
        ctx.set_profiling(True)
        if mode == "spmd":
            q.comm.timing(on=1)
    try:
        _pr = torch.cuda.get_device_properties(local_rank)
        _pci = "%04x:%02x:%02x.0" % (_pr.pci_domain_id, _pr.pci_bus_id, _pr.pci_device_id)
    except Exception:
        _pci = None
    tele = GpuTelemetry(local_rank, pci=_pci)
    tele.start()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    telemetry = tele.result()
    progress(f"{args.steps} timed steps done")
    per_rank = comm_t = None
    if mode == "mg":
        per_rank = [mg.stats(r) for r in range(mg.ndev)]
        comm_t = [mg.comm_timing(r, on=0) for r in range(mg.ndev)]
        mg.set_profiling(False)
        st = per_rank[0]
        panel_counts = [sum(s_["panels_fast"] for s_ in per_rank), sum(s_["panels_fallback"] for s_ in per_rank)]
    else:
        st = ctx.stats()
        ctx.set_profiling(False)
        panel_counts = list(ctx.panel_counters())
        if mode == "spmd":
            comm_t = q.comm.timing(on=0)
    if spmd:
        t = torch.tensor([dt], dtype=torch.float64)  # host tensor: the control plane is gloo
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()

    resid = None
    if not args.no_residual:
        if mode == "single":
            H = pkg.DistributedHouseholderQRStruct(A, alpha)
            A0 = pkg.rand_colmajor(m, n, seed, dev)
            resid = pkg.residual(H, A0)
            del A0
        elif mode == "mg":
            resid = mg.residual(seed)
        else:
            resid = q.residual(seed)

    ms_step = dt / args.steps * 1e3
    value = flops_qr(m, n) / (dt / args.steps) / 1e9
    dom, rl_all = roofline_groups(st, args.steps, m, n, nb)

    if mode == "single":
        par = "single GPU"
    elif mode == "mg":
        par = (f"1-D block-cyclic column split x{mg.ndev}, one process / one host thread per GPU, panel broadcast: {mg.transport}"
               + ("" if world > 1 else f" ({args.logical_ranks} logical ranks on ONE GPU: development run)"))
    else:
        par = f"1-D block-cyclic column split x{world}, one process per GPU, RCCL panel broadcast (ncclCommInitRank)"
    out = {
        "metric": "QR GFLOP/s (F = 2mn^2 - 2/3 n^3), ||A-QR||/||A|| alongside",
        "value": value, "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{m}x{n} Float64 dense QR, " +
                               (f"blocked nb=128 (BASELINE configs[{2 if world == 1 else 3}])" if nb else
                                "unblocked rank-1 (BASELINE configs[1])"),
                   "m": m, "n": n, "nb": nb, "parallelism": par},
        "residual": resid,
        "roofline": ({k: dom[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "launches",
                                          "launches_per_step", "avg_launch_ms", "ms_per_step", "traffic_bytes_per_step",
                                          "traffic_counted_launches_per_step", "traffic_ratio_to_algorithmic", "traffic_GBps",
                                          "traffic_null_because") if k in dom} if dom else None),
        "roofline_all": rl_all,
        **({"reflector_apply": {
            # SURVEY 8(d)'s per-unit figure (16 B per element and reflector: one read + one write) against what the
            # launches move: the ratio is the number of reflectors a pass applies per load/store of a column
            "bytes_one_reflector_per_pass": 16.0 * sum((m - j) * (n - j - 1) for j in range(n)) * args.steps,
            "bytes_as_implemented": st["bytes_rank1"],
            "reflectors_per_pass": 16.0 * sum((m - j) * (n - j - 1) for j in range(n)) * args.steps / st["bytes_rank1"],
            "equivalent_GBps_at_16B_per_element_and_reflector":
                16.0 * sum((m - j) * (n - j - 1) for j in range(n)) * args.steps / st["ms_rank1"] / 1e6}}
           if (not nb and st["bytes_rank1"] > 0) else {}),
        "traffic_note": "roofline.traffic = HBM bytes (read + write) per timed launch group -- the SAME launches `launches` and "
                        "`avg_launch_ms` count -- = traffic_bytes_per_step / launches_per_step; traffic_bytes_per_step = the bytes of "
                        "one factorisation from the committed rocprofv3 --pmc run of the same kernel symbol on the same workload "
                        "(FETCH_SIZE x2 on gfx950, WRITE_SIZE; separate passes over the torch-free driver; its "
                        "traffic_counted_launches_per_step wide launches carry the group's bytes, the narrow look-ahead launches "
                        "of the template are not counted), traffic_GBps = traffic_bytes_per_step / ms_per_step; used only while "
                        "the kernel sources still have the git blob ids stamped into profiles/pmc_traffic_current.json, "
                        "otherwise null with the reason",
        "gpu_telemetry": telemetry,
        **attempt_fields(attempt),
        "phase_ms_per_step": {k: st[k] / args.steps for k in st if k.startswith("ms_") and st[k] > 0},
        "panels_fast_fallback": panel_counts,
    }
    if mode == "mg" and mg.transport == "rccl":
        out["bcast_tuning"] = mg.bcast_tuning()
        out["rccl_nranks"] = mg.rccl_nranks()
    elif mode == "spmd":
        out["bcast_tuning"] = q.comm.bcast_tuning()
        out["rccl_nranks"] = q.comm.rccl_nranks()
    if mode != "single":
        out["roofline_note"] = "per-GPU figures of rank 0 (every rank runs the same kernels on 1/N of the columns)"
    # per-rank diagnosis of a multi-GPU run: where each rank's step went (ms per step) -- the wide stream's GEMMs, the panel
    # chain (lane busy time), the collectives (device time incl. the wait for the peers), and what is left of the step
    # beside the wide stream's kernels (waiting for the lane or a broadcast)
    def _diag(s_, t_):
        wide = (s_["ms_gemm_avw"] + s_["ms_gemm_vta"] + s_["ms_gemm_tw"]) / args.steps
        return {"wide_gemm_ms": round(wide, 2), "panel_chain_ms": round(s_["ms_panel"] / args.steps, 2),
                "bcast_ms": round(t_["bcast_ms"] / args.steps, 2), "n_bcast": t_["n_bcast"] / args.steps,
                "allreduce_ms": round(t_["allreduce_ms"] / args.steps, 2),
                "wide_stream_not_in_gemms_ms": round(ms_step - wide, 2)}
    if mode == "mg" and mg.ndev > 1:
        out["per_rank_ms_per_step"] = [_diag(per_rank[r], comm_t[r]) for r in range(mg.ndev)]
    elif mode == "spmd":
        mine = _diag(st, comm_t)
        allr = [None] * world
        dist.all_gather_object(allr, mine)
        out["per_rank_ms_per_step"] = allr
    if per_rank is not None:
        out["bcast_bytes_per_step_per_rank"] = per_rank[0]["bytes_bcast"] / args.steps
        out["per_rank_gemm_ms_per_step"] = [round((s_["ms_gemm_avw"] + s_["ms_gemm_vta"] + s_["ms_gemm_tw"]) / args.steps, 2)
                                            for s_ in per_rank]
        out["per_rank_panel_ms_per_step"] = [round(s_["ms_panel"] / args.steps, 2) for s_ in per_rank]
    if rank == 0 and mode == "single":
        try:
            import ctypes as _ct
            B, bh = pkg.bench_context(local_rank)  # libdhqr_bench.so: the micro-benchmarks are not in the product library
            o4 = (_ct.c_double * 4)()   # 4 waves/SIMD, accumulators in VGPRs: the achievable issue rate
            pkg.bench_check(B, B.dhqr_bench_issue2_f64(bh, 0, 1024, 256, o4))
            out["fp64_mfma_ubench_tflops"] = o4[2]
            out["fp64_mfma_ubench_note"] = ("v_mfma_f64_16x16x4_f64 only, 4 waves/SIMD, VGPR accumulators "
                                            "(16 AGPR accumulators per wave issue at half rate: %.1f TFLOP/s)"
                                            % pkg.bench_mfma_tflops(local_rank))
            out["stream_ubench_gbps"] = pkg.bench_stream_gbps(1 << 30, local_rank)
            # the wide kernels alone on synthetic operands of the first (largest) update + the shader clock under them
            g4 = (_ct.c_double * 4)()
            iso = {}
            for kind, name in ((2, "k_gemm_nn_quad K=512"), (0, "k_gemm_nn_sub K=256"), (1, "k_gemm_tn2 + its split-K reduction")):
                pkg.bench_check(B, B.dhqr_bench_gemm_f64(bh, kind, 16384, 16384, 3, g4))
                iso[name] = {"tflops": g4[1], "frac_of_peak": g4[1] / PEAK_FP64_MFMA_TFLOPS, "shader_mhz": g4[2]}
            out["gemm_kernels_in_isolation_16384"] = iso
        except Exception as e:  # diagnostics only
            out["ubench_error"] = repr(e)
        del A
        torch.cuda.empty_cache()
        if not args.no_also and args.config == "blocked" and not args.n and not args.m:
            # the other single-GPU configurations of BASELINE.json, driver-timed in the same run (a few seconds each)
            out["also"] = []
            for what, fn in (("unblocked 8192^2", lambda: also_unblocked(pkg, torch, ctx, dev)),
                             ("row split 262144x4096", lambda: also_tallskinny(pkg, torch)),
                             ("solve", lambda: also_solve(pkg, torch, dev, lapack=not args.no_cpu_baseline)),
                             ("blocked 8192^2 / 16384^2", lambda: also_sizes(pkg, torch, ctx, dev))):
                try:
                    progress(f"also: {what}")
                    out["also"].append(fn())
                except Exception as e:  # never takes the headline line down
                    out["also"].append({"config": {"workload": what}, "error": repr(e)[:300]})
                torch.cuda.empty_cache()
            try:
                progress("host-in / host-out call")
                out["host_in_out"] = host_in_out(pkg, torch, dev, m, n)
                out["host_in_out_ms"] = out["host_in_out"]["ms"]
            except Exception as e:
                out["host_in_out"] = {"error": repr(e)[:300]}
            torch.cuda.empty_cache()
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(m, n)
            out["cpu_baseline"]["distributed_structure"] = cpu_baseline_distributed()
            out["cpu_baseline"]["lapack_dgeqrf"] = cpu_baseline_lapack()
            out["host_cores"] = os.cpu_count()
    emit(out, rank)
    if mg is not None:
        mg.close()
    if spmd:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
