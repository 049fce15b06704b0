"""Developer timing probe (not the contract bench): per-phase device times of the factorisation."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import __graft_entry__ as g  # noqa: E402

pkg = g.import_package()


def run(n, nb, m=None, reps=2, resid=True):
    m = m or n
    ctx = pkg.get_context(0)
    out = {"m": m, "n": n, "nb": nb}
    for r in range(reps):
        A = pkg.rand_colmajor(m, n, 0, "cuda:0")
        al = torch.zeros(n, dtype=torch.float64, device="cuda:0")
        torch.cuda.synchronize()
        ctx.reset_stats()
        ctx.set_profiling(r == reps - 1)
        t0 = time.perf_counter()
        pkg.householder_(A, al, nb=nb)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out[f"t{r}"] = dt
    st = ctx.stats()
    ctx.set_profiling(False)
    flops = 2.0 * m * n * n - 2.0 / 3.0 * n ** 3
    out["gflops"] = flops / dt / 1e9
    out["stats"] = {k: (round(v, 3) if isinstance(v, float) else v) for k, v in st.items() if v}
    if st["ms_gemm_avw"]:
        out["avw_tflops"] = st["flops_gemm_avw"] / st["ms_gemm_avw"] / 1e9
        out["vta_tflops"] = st["flops_gemm_vta"] / st["ms_gemm_vta"] / 1e9
    if st["ms_rank1"]:
        out["rank1_GBps"] = st["bytes_rank1"] / st["ms_rank1"] / 1e6
    if resid:
        H = pkg.DistributedHouseholderQRStruct(A, al)
        A0 = pkg.rand_colmajor(m, n, 0, "cuda:0")
        out["resid"] = pkg.residual(H, A0)
    print(json.dumps(out), flush=True)
    return out


if __name__ == "__main__":
    print("mfma f64 ubench TF/s:", pkg.bench_mfma_tflops(0), " stream GB/s:", pkg.bench_stream_gbps(1 << 30, 0), flush=True)
    cfgs = [(2048, 0), (2048, 128), (8192, 0), (8192, 128), (16384, 128)]
    if len(sys.argv) > 1:
        cfgs = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]]
    for cfg in cfgs:  # n,nb or n,nb,m
        n, nb = cfg[0], cfg[1]
        try:
            run(n, nb, m=cfg[2] if len(cfg) > 2 else None)
        except Exception as e:  # keep going: this is a diagnostic
            print("FAILED", n, nb, repr(e), flush=True)
