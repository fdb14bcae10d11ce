"""Does RCCL accept two ranks on ONE device?  (If so the native RCCL loop can be exercised with N > 1 on a 1-GPU box.)"""
import os
import torch
import torch.distributed as dist

rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
t = torch.full((4,), float(rank + 1), device="cuda:0", dtype=torch.float64)
dist.all_reduce(t)
torch.cuda.synchronize()
print("rank", rank, "allreduce ->", t.tolist(), flush=True)
dist.destroy_process_group()
