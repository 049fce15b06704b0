"""Does a stream of SMALL kernels on a second stream slow the wide trailing-update GEMMs down by more than the resources
it uses?  Runs dhqr_bench_gemm_f64 (kind 2 = k_gemm_nn_quad K = 512, kind 1 = k_gemm_tn2) on a 32768 x 32768 operand alone,
then with a background thread launching one-workgroup kernels (1 element += 1) on another stream at several rates.
Every kernel boundary carries an acquire / release of the caches; this probe measures what that costs the GEMM next door
(the reason the look-ahead lane cannot simply be made to run concurrently: profiles/r03_panel_server.txt)."""
import ctypes
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import __graft_entry__ as g  # noqa: E402


def main():
    pkg = g.import_package()
    L, bh = pkg.bench_context(0)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
    reps = 12
    s2 = torch.cuda.Stream()
    y = torch.zeros(64, device="cuda:0")
    big = torch.zeros(1 << 22, device="cuda:0")  # 16 MiB: a kernel that also moves some data

    def gemm(kind):
        out = (ctypes.c_double * 4)()
        pkg.bench_check(L, L.dhqr_bench_gemm_f64(bh, kind, n, n, reps, out))
        return out[0], out[1]

    for kind, name in ((2, "k_gemm_nn_quad K=512"), (1, "k_gemm_tn2")):
        ms0, tf0 = gemm(kind)
        print(f"{name} alone: {ms0:.3f} ms/launch {tf0:.2f} TFLOP/s", flush=True)
        for label, pause, tensor in (("tiny kernels, as fast as the host launches", 0.0, y), ("tiny kernels, ~10 per ms", 1e-4, y),
                                     ("tiny kernels, ~2 per ms", 5e-4, y), ("16 MiB kernels, ~2 per ms", 5e-4, big)):
            stop = threading.Event()
            count = [0]

            def noise():
                with torch.cuda.stream(s2):
                    while not stop.is_set():
                        tensor.add_(1.0)
                        count[0] += 1
                        if pause:
                            time.sleep(pause)
                        elif count[0] % 256 == 0:
                            s2.synchronize()  # keep the queue bounded

            th = threading.Thread(target=noise)
            th.start()
            time.sleep(0.05)
            c0, t0 = count[0], time.perf_counter()
            ms, tf = gemm(kind)
            dt = time.perf_counter() - t0
            rate = (count[0] - c0) / dt / 1e3
            stop.set()
            th.join()
            torch.cuda.synchronize()
            print(f"  + {label}: {ms:.3f} ms/launch ({ms / ms0:.3f} x), {rate:.1f} launches per ms next door", flush=True)


if __name__ == "__main__":
    main()
