"""A solve whose persistent Q'b kernel cannot get all its workgroups resident (most compute units held by another stream's
kernel: libdhqr_bench.so's dhqr_debug_hold_cus) must be REPEATED with the per-step kernels by the first synchronising call
and return the right x (csrc/dhqr_api.hip: pipe_error_check).  Run in a process of its own (tests/test_gpu_kernels.py starts
it): with the dozens of streams a long pytest session has created, the holder's stream may share a hardware queue with the
solve's, and the two simply run one after the other.  Prints one JSON line per fill level."""
import json
import os, sys, time, ctypes
os.environ["DHQR_SOLVE_PIPE"]="3"; os.environ["DHQR_TUNE"]="spin_limit=20000"; os.environ["DHQR_SMALL"]="0"
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT","/root/repo"))
import torch, __graft_entry__ as g
pkg=g.import_package(); BL=pkg._lib.lib_bench(); P=ctypes.c_void_p
h=P(); assert BL.dhqr_create(ctypes.byref(h),0)==0
m=n=8192
A=pkg.rand_colmajor(m,n,7,"cuda:0"); al=torch.zeros(n,dtype=torch.float64,device="cuda:0")
assert BL.dhqr_factor_f64(h,P(A.data_ptr()),m,n,m,P(al.data_ptr()),128)==0
b=pkg.rand_vector_device(m,8,"cuda:0"); good=b.clone()
assert BL.dhqr_solve_f64(h,P(A.data_ptr()),m,n,m,P(al.data_ptr()),P(good.data_ptr()))==0 and BL.dhqr_synchronize(h)==0
for held in (160,200,224,240):
    for wait in (0.2,):
        t0=time.perf_counter()
        assert BL.dhqr_debug_hold_cus(h,held,1500,0)==0
        time.sleep(0.05)
        bb=b.clone()
        assert BL.dhqr_solve_f64(h,P(A.data_ptr()),m,n,m,P(al.data_ptr()),P(bb.data_ptr()))==0
        time.sleep(wait)
        t1=time.perf_counter(); BL.dhqr_debug_hold_cus(h,0,0,1); t2=time.perf_counter()
        rc=BL.dhqr_synchronize(h); t3=time.perf_counter()
        rr=ctypes.c_int64(); BL.dhqr_get_solve_retries(h,ctypes.byref(rr))
        err=float((bb[:n]-good[:n]).abs().max())
        print(json.dumps({"held": held, "wait_s": wait, "sync_ms": round((t3 - t2) * 1e3, 2), "rc": rc, "retries": rr.value,
                          "max_abs_diff_to_undisturbed_x": err, "max_abs_x": float(good[:n].abs().max())}), flush=True)
