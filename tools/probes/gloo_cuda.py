"""Probe (GPU box): which torch.distributed gloo collectives work on device tensors with two ranks on one GPU."""
import os, sys, socket
import torch, torch.distributed as dist, torch.multiprocessing as mp
def w(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    t = torch.full((4,), float(rank + 1), device=dev)
    dist.broadcast(t, src=0); torch.cuda.synchronize()
    print(rank, "broadcast contiguous ->", t.tolist(), flush=True)
    u = torch.full((2, 3, 2, 2), float(rank + 1), device=dev).contiguous(memory_format=torch.channels_last)
    dist.broadcast(u, src=0); torch.cuda.synchronize()
    print(rank, "broadcast channels_last ->", u.flatten()[:3].tolist(), flush=True)
    v = torch.full((2, 3, 2, 2), float(rank + 1), device=dev).contiguous(memory_format=torch.channels_last)
    f = torch.as_strided(v, (v.numel(),), (1,), v.storage_offset())
    dist.broadcast(f, src=0); torch.cuda.synchronize()
    print(rank, "broadcast flat alias ->", v.flatten()[:3].tolist(), flush=True)
    a = torch.full((4,), float(rank + 1), device=dev)
    dist.all_reduce(a); torch.cuda.synchronize()
    print(rank, "all_reduce ->", a.tolist(), flush=True)
    o = torch.empty(8, device=dev)
    dist.all_gather_into_tensor(o, torch.full((4,), float(rank + 1), device=dev)); torch.cuda.synchronize()
    print(rank, "all_gather_into_tensor ->", o.tolist(), flush=True)
    dist.destroy_process_group()
if __name__ == "__main__":
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(w, args=(2, port), nprocs=2, join=True)
