"""ComplexF64 column split (dhqr_zdist.h) on one GPU: the SPMD driver at one rank next to dhqr_factor_c64_nb (same schedule,
device-resident), and qr!(A; ndev) host-in / host-out over 1 / 2 / 4 logical ranks sharing the GPU (peer-copy transport)."""
import ctypes, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from importlib import import_module
import __graft_entry__
pkg = __graft_entry__.import_package()
D = import_module("dhqr_amd.distributed")
api = import_module("dhqr_amd.api")
L = import_module("dhqr_amd._lib").lib()
dev = "cuda:0"
ctx = api.get_context(0)
comm = D.Communicator.from_torch(ctx)
for n in (4096, 8192):
    m = n
    A = pkg.rand_colmajor_c(m, n, 3, dev)
    A0 = A.clone()
    al = torch.zeros(n, dtype=torch.complex128, device=dev)
    def run_nb():
        A.copy_(A0); torch.cuda.synchronize(); t = time.perf_counter()
        pkg.qr_(A, nb=64); torch.cuda.synchronize(); return (time.perf_counter() - t) * 1e3
    def run_cs():
        A.copy_(A0); torch.cuda.synchronize(); comm.order_with_torch(None) if False else None
        t = time.perf_counter()
        rc = L.dhqr_cs_factor_c64(comm.handle, ctypes.c_void_p(A.data_ptr()), m, n, A.stride(1), ctypes.c_void_p(al.data_ptr()))
        assert rc == 0, L.dhqr_last_error()
        ctx.synchronize() if hasattr(ctx, "synchronize") else None
        torch.cuda.synchronize(); return (time.perf_counter() - t) * 1e3
    for f, name in ((run_nb, "dhqr_factor_c64_nb"), (run_cs, "dhqr_cs_factor_c64 (1 rank)")):
        f(); ts = [f() for _ in range(3)]
        print(f"{n} x {n} ComplexF64 {name}: {min(ts):.1f} ms", flush=True)
    H1 = A.cpu().numpy().copy()
    A.copy_(A0); pkg.qr_(A, nb=64); torch.cuda.synchronize()
    print("   max |H_cs - H_nb| / max|H| =", np.abs(H1 - A.cpu().numpy()).max() / np.abs(H1).max(), flush=True)
    del A, A0
for ranks in (1, 2, 4):
    mg = pkg.MultiGpuQR(devices=[0] * ranks)
    n = m = 4096
    A0 = np.asfortranarray((np.random.default_rng(1).random((m, n)) - 0.5) + 1j * (np.random.default_rng(2).random((m, n)) - 0.5))
    A = A0.copy(order="F"); mg.qr_(A)
    ts = []
    for _ in range(2):
        A = A0.copy(order="F"); t = time.perf_counter(); H, al = mg.qr_(A); ts.append((time.perf_counter() - t) * 1e3)
    x = np.random.default_rng(3).random(n) + 0j
    b = A0 @ x
    xs = pkg.ldiv(pkg.DistributedHouseholderQRStruct(H, al), b)
    print(f"qr!(A; ndev = {ranks} logical ranks on one GPU) 4096^2 ComplexF64 host in / out: {min(ts):.1f} ms, |x - x0|max {np.abs(np.asarray(xs) - x).max():.2e}", flush=True)
    mg.close()
