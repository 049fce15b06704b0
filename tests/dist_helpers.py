"""Test-only helpers: gloo rank spawning, the EMULATED library (tests/simt: csrc/ host-compiled for the CPU),
gloo-backed communicator callbacks for it, and numpy stand-ins for the row-split backend.
The product never imports this file."""
import os
import sys
import traceback

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NB = 128


def run_ranks(fn, world_size, *args):
    """spawn `world_size` gloo ranks running fn(rank, world_size, *args); re-raise failures"""
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_entry, args=(fn, r, world_size, port, q, args)) for r in range(world_size)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    errs = [r for r in results if r[1] is not None]
    if errs:
        raise AssertionError("rank failures:\n" + "\n".join(f"[rank {r[0]}] {r[1]}" for r in errs))
    return {r[0]: r[2] for r in results}


def _entry(fn, rank, world_size, port, q, args):
    try:
        if ROOT not in sys.path:
            sys.path.insert(0, ROOT)
        import torch.distributed as dist
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        os.environ["OMP_NUM_THREADS"] = "2"
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("gloo", rank=rank, world_size=world_size)
        out = fn(rank, world_size, *args)
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, None, out))
    except BaseException:
        q.put((rank, traceback.format_exc(), None))


# ---------------------------------------------------------------------------------------------------
# The product's HipBackend driven by the EMULATED library (tests/simt: the unmodified csrc/ compiled for
# the CPU): the same marshalling code, the same C entry points, CPU tensors instead of device tensors.
# Lets the world_size-2/3 gloo tests run the real panel_factor / panel_apply / form_r0 / backsub kernels.
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def build_emulated_library(outdir):
    """host-compile the library's translation units (csrc/dhqr_api.hip, csrc/dhqr_unblocked.hip) against tests/simt/fake
    (fiber mode) -> path of the .so"""
    import subprocess
    so = os.path.join(str(outdir), "libdhqr_emulated.so")
    csrc = os.path.join(ROOT, "distributedhouseholderqr.jl_amd", "csrc")
    subprocess.check_call([CLANG, "-x", "c++", "-std=c++20", "-O2", "-DSIMT_FIBERS", "-fPIC", "-shared",
                           "-Wno-unknown-attributes", "-Wno-psabi", "-Wno-unused-value",
                           "-I", os.path.join(ROOT, "tests", "simt", "fake"),
                           os.path.join(csrc, "dhqr_api.hip"), os.path.join(csrc, "dhqr_unblocked.hip"), "-o", so])
    return so


def load_emulated_library(so):
    import ctypes
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "dhqr_lib_signatures", os.path.join(ROOT, "distributedhouseholderqr.jl_amd", "_lib.py"))
    sig = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sig)
    L = ctypes.CDLL(so)
    for name, (res, args) in sig.SIGNATURES.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    L.Stats = sig.Stats  # the dhqr_stats structure class the prototypes were declared with
    return L


def emulated_rank(so, nranks, rank):
    """(library, ctx handle, Communicator) of one gloo rank on the EMULATED library: the product's SPMD drivers
    (dhqr_cs_*) run unmodified; broadcast / all-reduce go through the CALLBACK transport into torch.distributed
    (gloo) on the numpy-backed "device" memory."""
    import ctypes
    import importlib
    import torch.distributed as dist
    import __graft_entry__ as g
    g.import_package()
    D = importlib.import_module("dhqr_amd.distributed")
    L = load_emulated_library(so)
    h = ctypes.c_void_p()
    assert L.dhqr_create(ctypes.byref(h), 0) == 0

    def _view(ptr, count):
        return torch.from_numpy(np.ctypeslib.as_array((ctypes.c_double * count).from_address(ptr)))

    def bcast(ptr, nbytes, root):
        dist.broadcast(_view(ptr, nbytes // 8), src=root)

    def allreduce(ptr, count):
        dist.all_reduce(_view(ptr, count))

    comm = D.Communicator.from_callbacks(h, L, nranks, rank, bcast, allreduce)
    return L, h, comm, D


# ---------------------------------------------------------------------------------------------------
# P logical ranks on ONE GPU, as P Python threads of one process, each with its own dhqr_ctx and a CALLBACK
# communicator whose broadcast / all-reduce are device-to-device copies on that GPU: the SPMD entry points a
# Julia worker binds (dhqr_cs_qr_darray_*, dhqr_cs_ldiv_darray_*, dhqr_cs_*) run with P > 1 on the 1-GPU box.
# (ctypes releases the GIL for the duration of a C call and re-acquires it inside the callbacks.)
def gpu_thread_ranks(P, fn, timeout=600):
    """run fn(rank, comm, lib) on P rank threads sharing cuda:0; returns {rank: result}; re-raises failures"""
    import ctypes
    import importlib
    import threading
    import __graft_entry__ as g
    pkg = g.import_package()
    D = importlib.import_module("dhqr_amd.distributed")
    L = pkg._lib.lib()
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    hip.hipMemcpy.restype = ctypes.c_int
    hip.hipDeviceSynchronize.restype = ctypes.c_int
    H2D, D2H, D2D = 1, 2, 3
    bar = threading.Barrier(P, timeout=timeout)
    shared = {"src": None, "parts": [None] * P}
    out, errs = {}, {}

    def run(rank):
        try:
            assert hip.hipSetDevice(0) == 0
            ctx = pkg.Context(0)

            def bcast(ptr, nbytes, root):
                if rank == root:
                    shared["src"] = ptr
                bar.wait()
                if rank != root and nbytes:
                    assert hip.hipMemcpy(ptr, shared["src"], nbytes, D2D) == 0
                    assert hip.hipDeviceSynchronize() == 0
                bar.wait()

            def allreduce(ptr, count):
                part = np.zeros(count)
                assert hip.hipMemcpy(part.ctypes.data, ptr, 8 * count, D2H) == 0
                shared["parts"][rank] = part
                bar.wait()
                tot = np.zeros(count)
                for p in shared["parts"]:  # the same order on every rank: identical sums
                    tot += p
                assert hip.hipMemcpy(ptr, tot.ctypes.data, 8 * count, H2D) == 0
                assert hip.hipDeviceSynchronize() == 0
                bar.wait()

            comm = D.Communicator.from_callbacks(ctx.handle, L, P, rank, bcast, allreduce)
            try:
                out[rank] = fn(rank, comm, L)
            finally:
                comm.close()
                ctx.close()
        except BaseException:
            errs[rank] = traceback.format_exc()
            bar.abort()

    ts = [threading.Thread(target=run, args=(r,)) for r in range(P)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout)
    if errs:
        raise AssertionError("rank failures:\n" + "\n".join(f"[rank {r}] {e}" for r, e in sorted(errs.items())))
    assert len(out) == P, "a rank thread did not finish"
    return out
