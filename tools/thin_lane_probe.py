"""Could the look-ahead lane's kernels run IN the slots a retiring k_gemm_nn_quad workgroup frees, beside the wide launch,
and what would that cost?  dhqr_bench_lane_probe_f64 (one context, its high-priority stream) runs alone and while
dhqr_bench_gemm_f64 (kind 2: k_gemm_nn_quad on a 32768^2 operand, another context) runs: Gram products of a 32768 x 128
panel in 234 / 64 / 32 / 16 workgroups of the subtraction's footprint, and a one-workgroup stand-in for the panel kernels
with 80 KB (fits a freed slot) or 136 KB (needs an empty CU, like k_panel_top today) of LDS.
  python tools/thin_lane_probe.py [n=32768]"""
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
    if len(sys.argv) > 2 and sys.argv[2] == "null":  # the subtraction on the NULL stream (what bench.py's single-GPU run uses)
        pkg.bench_check(L, L.dhqr_set_stream(bh, None))
        print("the subtraction runs on the NULL stream", flush=True)
    h2 = ctypes.c_void_p()
    pkg.bench_check(L, L.dhqr_create(ctypes.byref(h2), 0))

    def gemm(reps):
        out = (ctypes.c_double * 4)()
        pkg.bench_check(L, L.dhqr_bench_gemm_f64(bh, 2, n, n, reps, out))
        return out[0], out[1]

    def lane(nsplit, lds, reps):
        out = (ctypes.c_double * 4)()
        pkg.bench_check(L, L.dhqr_bench_lane_probe_f64(h2, n, nsplit, lds, reps, out))
        return out[0], out[1], out[2]

    ms0, tf0 = gemm(8)
    print(f"k_gemm_nn_quad alone: {ms0:.3f} ms/launch {tf0:.2f} TFLOP/s", flush=True)
    for nsplit in (234, 32):
        for lds in (0, 80, 136, -136):  # 0: the REAL k_panel_top + k_build_t; negative: the register-heavy stand-in (128 VGPRs x 1024 threads = a whole CU)
            a = lane(nsplit, lds, 20)
            res = {}
            stop = threading.Event()

            def bg():
                tot, k = [0.0, 0.0, 0.0], 0
                while not stop.is_set():
                    r = lane(nsplit, lds, 4)
                    tot = [x + y for x, y in zip(tot, r)]
                    k += 1
                res["lane"] = [x / max(k, 1) for x in tot]
                res["reps"] = 4 * k

            th = threading.Thread(target=bg)
            th.start()
            time.sleep(0.3)
            t0 = time.perf_counter()
            ms, tf = gemm(12)
            dt = time.perf_counter() - t0
            stop.set()
            th.join()
            b = res["lane"]
            regs = "REAL k_panel_top + k_build_t:" if lds == 0 else ("128" if lds < 0 else "40")
            print(f"Gram in {nsplit:3d} workgroups, stand-in {regs} VGPRs, {abs(lds):3d} KB LDS:  alone gram {a[0] * 1e3:7.1f} us, stand-in {a[1] * 1e3:7.1f} us"
                  f" | beside the subtraction gram {b[0] * 1e3:8.1f} us, stand-in {b[1] * 1e3:8.1f} us ({res['reps']} repetitions)"
                  f" | subtraction {ms:.3f} ms/launch = {ms / ms0:.3f} x", flush=True)
    L.dhqr_destroy(h2)


if __name__ == "__main__":
    main()
