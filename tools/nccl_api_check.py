"""developer check on a 1-GPU box: the RCCL process group accepts the exact calls bench.py makes at
N > 1 (uint8 gather with async_op, float64 all_reduce SUM/MAX, barrier) -- with world_size 1."""
import os
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
mine = torch.arange(2 * 56 * 8 * 4, dtype=torch.uint8, device=dev).reshape(2, 56, 8, 4)
recv = [torch.empty_like(mine)]
w = dist.gather(mine, gather_list=recv, dst=0, async_op=True)
w.wait()
torch.cuda.synchronize()
assert torch.equal(recv[0], mine)
t = torch.tensor([1.5, 2.5], dtype=torch.float64, device=dev)
dist.all_reduce(t); dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
print("nccl api ok", t.tolist())
dist.destroy_process_group()
