"""Can the C-ABI's RCCL all-gather (racc_hip_allgather_results) be rehearsed with TWO ranks on the ONE GPU of a test box?
Two engine contexts on device 0, one communicator rank each (two host threads: ncclCommInitRank blocks until every rank has
called it).  Prints what RCCL answers.  (Expected: ncclCommInitRank refuses two ranks on one device — "Duplicate GPU detected".)"""
import os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rayaccel_amd as ra

uid = ra.Comm.unique_id()
out = [None, None]


def rank(r):
    try:
        with ra.Context(device=0) as ctx:
            comm = ra.Comm(ctx, uid, r, 2)
            n = 1024
            send = ctx.alloc(n * 16); recv = ctx.alloc(2 * n * 16)
            send.upload(np.full(n * 4, r + 1, np.uint32))
            comm.allgather_results(send.ptr, recv.ptr, n)
            ctx.synchronize()
            got = recv.download(np.uint32, 2 * n * 4)
            out[r] = "ok: gathered %s" % np.unique(got).tolist()
            comm.destroy()
    except Exception as e:   # noqa: BLE001
        out[r] = "error: %s" % e


ts = [threading.Thread(target=rank, args=(r,)) for r in (0, 1)]
for t in ts: t.start()
for t in ts: t.join(timeout=120)
print("two ranks on GPU 0:", out)
