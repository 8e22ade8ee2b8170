import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("LOCAL_RANK", "0")
torch.cuda.set_device(0); dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)
import sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from online_gp_amd.distributed import allreduce_sum_
ts = [torch.ones(172 * 125000, device=dev), torch.ones(3, dtype=torch.float64, device=dev)]
allreduce_sum_(ts)  # world 1: returns early
h = [dist.all_reduce(t, async_op=True) for t in ts]; [x.wait() for x in h]
dist.barrier(); torch.cuda.synchronize()
print("nccl ok", float(ts[0][0]), dist.get_backend())
dist.destroy_process_group()
