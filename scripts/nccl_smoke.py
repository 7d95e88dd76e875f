import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
dev = torch.device("cuda", 0)
world, V, n = 1, 16, 1000
send = torch.arange(V * n, dtype=torch.int32, device=dev).reshape(V, n)
allb = torch.empty((world, V, n), dtype=torch.int32, device=dev)
parts = [allb[r] for r in range(world)]
comm = torch.cuda.Stream(dev)
ev = torch.cuda.Event(); ev.record(torch.cuda.current_stream())
with torch.cuda.stream(comm):
    comm.wait_event(ev)
    probe = torch.zeros((256,), dtype=torch.int32, device=dev)
    dist.gather(probe, [torch.empty_like(probe) for _ in range(world)], dst=0)
    dist.gather(send, parts, dst=0)
torch.cuda.synchronize()
dist.barrier()
t = torch.tensor([1.5], device=dev, dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX)
print("nccl gather ok:", bool(torch.equal(allb[0], send)), float(t.item()))
dist.destroy_process_group()
