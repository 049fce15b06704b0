"""TEST / BASELINE INFRASTRUCTURE: timing of the reference's DISTRIBUTED structure restated (oracle/dist_oracle.py:
contiguous column blocks, owners visited sequentially, one broadcast of the dense column Hj per column,
src/DistributedHouseholderQR.jl:115-148) with P processes x T threads on the host cores -- the CPU-baseline leg of
bench.py runs it (BASELINE configs[0]: 512 x 512, nprocs = 2; plus larger orders), nothing else does.

  python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 oracle/dist_bench.py 512 2048 4096
prints one JSON line on rank 0."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch.distributed as dist
    from oracle import dhqr_oracle as orc
    from oracle import dist_oracle as do
    dist.init_process_group("gloo")
    rank, P = dist.get_rank(), dist.get_world_size()
    out = []
    for n in [int(a) for a in sys.argv[1:]]:
        A = orc.rand_matrix(n, n, 0)
        lo, hi = do.column_blocks(n, P)[rank]
        Al = np.array(A[:, lo:hi], order="F", copy=True)
        alpha = np.zeros(n)
        del A
        dist.barrier()
        t0 = time.perf_counter()
        do.householder_darray(Al, n, n, alpha)
        dist.barrier()
        dt = time.perf_counter() - t0
        out.append({"n": n, "procs": P, "threads_per_proc": orc.num_threads(), "seconds": dt,
                    "gflops": (4.0 / 3.0) * n ** 3 / dt / 1e9})
    if rank == 0:
        print(json.dumps(out), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
