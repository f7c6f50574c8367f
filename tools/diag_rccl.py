"""diagnostic: host and device cost of a one-rank all-reduce through rccl.Communicator vs torch.distributed (nccl)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29612")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
from ga_ddpg_amd import rccl
comm = rccl.Communicator()
for n in (4, 2 * 1024 * 1024):
    x = torch.ones(n, dtype=torch.float32, device="cuda")
    for name, f in (("direct", lambda: comm.all_reduce_(x)), ("torch ", lambda: dist.all_reduce(x))):
        for _ in range(5):
            f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            f()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("%s n=%8d: host %.1f us/call, with drain %.1f us/call" % (name, n, (t1 - t0) / 200 * 1e6, (t2 - t0) / 200 * 1e6))
# does the call block the host while the stream is busy?
x = torch.ones(2 * 1024 * 1024, dtype=torch.float32, device="cuda")
for name, f in (("direct", lambda: comm.all_reduce_(x)), ("torch ", lambda: dist.all_reduce(x))):
    torch.cuda.synchronize()
    torch.cuda._sleep(int(2.4e9 * 0.02))            # ~20 ms of device work in front
    t0 = time.perf_counter()
    f()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print("%s behind 20 ms of queued work: host returns after %.2f ms (drain %.2f ms)" % (name, (t1 - t0) * 1e3, (time.perf_counter() - t0) * 1e3))
dist.destroy_process_group()
