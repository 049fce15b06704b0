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
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# the host driver only supports dmabuf IPC; must be in the environment before the HSA runtime starts
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

PEAK_FP64_MFMA_TFLOPS = 78.6   # AMD MI355X datasheet FP64 matrix (== vector) peak; the local
                               # MI355X_MICROARCH.md guide lists no FP64 number
PEAK_HBM_GBPS = 8000.0         # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


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


def tallskinny(args, pkg, torch, dist, world, rank, local_rank, dev, spmd):
    """BASELINE configs[4]: 262144 x 4096 Float64, rows split over the ranks (dhqr_rs_* / dhqr_mg_rs_*)."""
    m = args.m or 262144
    n = args.n or 4096
    mg = q = None
    if spmd:
        q = pkg.RowSplitQR(m, n, comm=pkg.Communicator.from_torch(pkg.get_context(local_rank)))
        ctxs = [pkg.get_context(local_rank)]

        def step():
            q.fill(0)
            q.factor()
    else:
        devices = list(range(world)) if (world > 1 or not args.logical_ranks) else [0] * args.logical_ranks
        mg = pkg.MultiGpuQR(devices=devices)
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

    for _ in range(args.warmup):
        step()
    barrier()
    if mg is not None:
        mg.reset_stats()
        mg.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    st = None
    if mg is not None:
        st = mg.stats(0)
        mg.set_profiling(False)
    if spmd:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
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


def pmc_traffic(kernel, m, n, nb, launches, work):
    """HBM bytes per launch of a roofline group from the committed counter runs (profiles/r02_pmc_traffic.json:
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over tools/pmc_driver, gfx950 x2 FETCH correction,
    calibrated on a streaming kernel).  Counters cannot be read from inside this process, so the figure is the one
    measured for the SAME kernel on the SAME workload; None when no matching measurement is committed."""
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", "r02_pmc_traffic.json")))
    except Exception:
        return None
    if nb and (m, n) == (32768, 32768):
        key = ("k_gemm_nn_sub<2,256>" if kernel.startswith("k_gemm_nn_sub") else
               ("k_gemm_tn2<2>" if kernel.startswith("k_gemm_tn") else None))
        e = pm.get("blocked32768_summary", {}).get(key) if key else None
        return e["bytes_per_launch"] if e else None
    if not nb and kernel.startswith("k_rankk"):
        # measured on the same kernel and workload (8192^2): HBM bytes / the algorithmic bytes of the launches as implemented
        e = pm.get("unblocked8192_k_rankk")
        return e["ratio"] * work / max(1, launches) if e else None
    return None


def roofline_groups(st, steps, m=0, n=0, nb=0):
    """per-kernel-group roofline entries from the hipEvent statistics of ONE rank"""
    groups = []
    if st["ms_gemm_avw"] > 0:
        # one timed group = ONE wide k_gemm_nn_sub launch on the caller's stream (rocprofv3 lists the narrow
        # look-ahead launches of the same template on the second stream as well: profiles/*_by_stream.csv)
        groups.append(dict(kernel="k_gemm_nn_sub (A -= V*W, FP64 MFMA)", bound="mfma", ms=st["ms_gemm_avw"],
                           launches=st["n_gemm_avw"], work=st["flops_gemm_avw"]))
        # one timed group = the TN launches of one wide update (two per two-panel update) + their split-K reductions
        groups.append(dict(kernel="k_gemm_tn2 / k_gemm_tn (W = [V_a V_b]'*A, FP64 MFMA)", bound="mfma", ms=st["ms_gemm_vta"],
                           launches=st["n_gemm_vta"], work=st["flops_gemm_vta"]))
    if st["ms_panel"] > 0:
        groups.append(dict(kernel="panel lane: Gram/Cholesky/replay/narrow-update kernels (dhqr_recon.h)", bound="hbm", ms=st["ms_panel"],
                           launches=st["n_panel"], work=st["bytes_panel"]))
    if st["ms_rank1"] > 0:
        # work = algorithmic HBM bytes of the launches AS IMPLEMENTED: a pass applies K reflectors to every trailing column
        # it loads and stores once (16 B per element and pass = 16/K B per element and reflector)
        groups.append(dict(kernel="k_rankk_fused (reflector apply, K reflectors per pass over the trailing columns)",
                           bound="hbm", ms=st["ms_rank1"], launches=st["n_rank1"], work=st["bytes_rank1"]))
    rl_all = []
    for gr in groups:
        if gr["bound"] == "mfma":
            ach, peak, unit = gr["work"] / gr["ms"] / 1e9, PEAK_FP64_MFMA_TFLOPS, "TFLOP/s"
        else:
            ach, peak, unit = gr["work"] / gr["ms"] / 1e6, PEAK_HBM_GBPS, "GB/s"
        rl_all.append({"kernel": gr["kernel"], "bound": gr["bound"], "achieved": ach, "peak": peak, "unit": unit,
                       "frac": ach / peak, "traffic": pmc_traffic(gr["kernel"], m, n, nb, gr["launches"], gr["work"]),
                       "launches": gr["launches"],
                       "avg_launch_ms": gr["ms"] / max(1, gr["launches"]), "total_ms": gr["ms"]})
    # the north star grades the trailing update: the DOMINANT (largest total time) MFMA group
    mf = [r for r in rl_all if r["bound"] == "mfma"]
    dom = max(mf, key=lambda r: r["total_ms"]) if mf else (max(rl_all, key=lambda r: r["total_ms"]) if rl_all else None)
    return dom, rl_all


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
    ap.add_argument("--logical-ranks", type=int, default=0,
                    help="development: run R ranks of the multi-GPU driver on ONE GPU (in-process peer-copy transport)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import __graft_entry__ as g
    pkg = g.import_package()

    # Launch modes (same C drivers underneath):
    #   python bench.py --gpus N                  ONE process drives N GPUs (dhqr_mg_*: a host thread per device, RCCL
    #                                             communicators from ncclCommInitAll)
    #   torchrun --nproc-per-node N bench.py ...  one process per GPU (dhqr_cs_* over ncclCommInitRank, the unique id
    #                                             shipped through torch.distributed)
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    spmd = env_world > 1
    if spmd and env_world != args.gpus:
        raise SystemExit(f"WORLD_SIZE={env_world} but --gpus {args.gpus}")
    world = args.gpus
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if spmd:
        dist.init_process_group("nccl", device_id=dev)

    if args.config == "tallskinny":
        return tallskinny(args, pkg, torch, dist, world, rank, local_rank, dev, spmd)
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

        def step():
            q.fill(seed)
            q.factor()
    elif world > 1 or args.logical_ranks:
        mode = "mg"
        devices = list(range(world)) if world > 1 else [0] * args.logical_ranks
        mg = pkg.MultiGpuQR(devices=devices)
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

    for _ in range(args.warmup):
        step()
    barrier()
    if mode == "mg":
        mg.reset_stats()
        mg.set_profiling(True)
    else:
        ctx.reset_stats()
        ctx.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    per_rank = None
    if mode == "mg":
        per_rank = [mg.stats(r) for r in range(mg.ndev)]
        mg.set_profiling(False)
        st = per_rank[0]
        panel_counts = [sum(s_["panels_fast"] for s_ in per_rank), sum(s_["panels_fallback"] for s_ in per_rank)]
    else:
        st = ctx.stats()
        ctx.set_profiling(False)
        panel_counts = list(ctx.panel_counters())
    if spmd:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
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
        "roofline": ({k: dom[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel",
                                          "launches", "avg_launch_ms")} if dom else None),
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
        "traffic_note": "HBM bytes per WIDE launch (read + write) from rocprofv3 --pmc FETCH_SIZE (x2 on gfx950) and WRITE_SIZE, separate "
                        "passes over the torch-free driver with the end-of-round kernels (tools/gpu_pmc3.sh, profiles/r02_pmc_traffic.json); "
                        "algorithmic C bytes per wide two-panel launch: 5.78 GB (NN: read + write) / 2.89 GB (TN: read) -> measured "
                        "8.97 / 3.64 GB",
        "phase_ms_per_step": {k: st[k] / args.steps for k in st if k.startswith("ms_") and st[k] > 0},
        "panels_fast_fallback": panel_counts,
    }
    if mode == "mg" and mg.transport == "rccl":
        out["bcast_tuning"] = mg.bcast_tuning()
    elif mode == "spmd":
        out["bcast_tuning"] = q.comm.bcast_tuning()
    if mode != "single":
        out["roofline_note"] = "per-GPU figures of rank 0 (every rank runs the same kernels on 1/N of the columns)"
    if per_rank is not None:
        out["bcast_bytes_per_step_per_rank"] = per_rank[0]["bytes_bcast"] / args.steps
        out["per_rank_gemm_ms_per_step"] = [round((s_["ms_gemm_avw"] + s_["ms_gemm_vta"] + s_["ms_gemm_tw"]) / args.steps, 2)
                                            for s_ in per_rank]
        out["per_rank_panel_ms_per_step"] = [round(s_["ms_panel"] / args.steps, 2) for s_ in per_rank]
    if rank == 0 and mode == "single":
        try:
            import ctypes as _ct
            o4 = (_ct.c_double * 4)()   # 4 waves/SIMD, accumulators in VGPRs: the achievable issue rate
            pkg._lib.check(pkg._lib.lib().dhqr_bench_issue2_f64(ctx.handle, 0, 1024, 256, o4))
            out["fp64_mfma_ubench_tflops"] = o4[2]
            out["fp64_mfma_ubench_note"] = ("v_mfma_f64_16x16x4_f64 only, 4 waves/SIMD, VGPR accumulators "
                                            "(16 AGPR accumulators per wave issue at half rate: %.1f TFLOP/s)"
                                            % pkg.bench_mfma_tflops(local_rank))
            out["stream_ubench_gbps"] = pkg.bench_stream_gbps(1 << 30, local_rank)
            # the wide kernels alone on synthetic operands of the first (largest) update + the shader clock under them
            g4 = (_ct.c_double * 4)()
            iso = {}
            for kind, name in ((0, "k_gemm_nn_sub K=256"), (1, "k_gemm_tn2")):
                pkg._lib.check(pkg._lib.lib().dhqr_bench_gemm_f64(ctx.handle, kind, 16384, 16384, 3, g4))
                iso[name] = {"tflops": g4[1], "frac_of_peak": g4[1] / PEAK_FP64_MFMA_TFLOPS, "shader_mhz": g4[2]}
            out["gemm_kernels_in_isolation_16384"] = iso
            out["power_note"] = ("rocm-smi while the kernels run (profiles/r02_power_telemetry.txt): MFMA-only loop 924 W, "
                                 "k_gemm_tn2 1164 W, k_gemm_nn_sub 1245 W (peaks 1327 W) of the 1400 W cap at 2.38-2.39 GHz")
        except Exception as e:  # diagnostics only
            out["ubench_error"] = repr(e)
        if not args.no_cpu_baseline:
            del A
            torch.cuda.empty_cache()
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
