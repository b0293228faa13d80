#!/usr/bin/env python
"""Probe (VERDICT r5 #9a): does RCCL accept TWO ranks on ONE device?  Expected: refused (a communicator wants one GPU per rank) —
the exact error text is recorded so that the first multi-GPU log can be triaged against it.  Every step runs under a short timeout;
the parent kills the group if a rank hangs.

    timeout 180 python tools/probes/rccl_two_ranks_one_device.py        (GPU box; prints one line per rank and a verdict)
"""
import datetime
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_DEBUG=os.environ.get("NCCL_DEBUG", "WARN"))
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    stage = "init_process_group"
    try:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=datetime.timedelta(seconds=45))
        stage = "all_reduce"
        t = torch.ones(1024, device=dev) * (rank + 1)
        dist.all_reduce(t)
        torch.cuda.synchronize()
        ret[rank] = "rank %d: ACCEPTED — all_reduce over 2 ranks on one device returned %s (expected %s)" % (rank, t[0].item(), 3.0)
        dist.destroy_process_group()
    except Exception as e:       # noqa: BLE001 — the text IS the result
        ret[rank] = "rank %d: REFUSED at %s: %s: %s" % (rank, stage, type(e).__name__, " ".join(str(e).split())[:600])
        os._exit(0)          # a half-made communicator can hang the interpreter's teardown: the answer is already with the parent


if __name__ == "__main__":
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mgr = mp.Manager()
    ret = mgr.dict()
    ctx = mp.spawn(worker, args=(2, port, ret), nprocs=2, join=False)
    ok = ctx.join(timeout=120)
    if not ok:
        for p in ctx.processes:
            if p.is_alive():
                p.kill()
        print("verdict: HUNG (no answer within 120 s; ranks killed) — partial:", dict(ret))
        sys.exit(0)
    for r in range(2):
        print(ret.get(r, "rank %d: no result" % r))
    print("verdict:", "RCCL accepts 2 ranks on one device" if all("ACCEPTED" in ret.get(r, "") for r in range(2)) else
          "RCCL refuses 2 ranks on one device (one GPU per rank): the 2-rank tests of this box use gloo on device tensors")
