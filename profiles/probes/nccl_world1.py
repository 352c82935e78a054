import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda",0))
import sys; sys.path.insert(0,'.')
from rocket_path_amd import sharding
t=torch.tensor([3.0,4.0,5.0,6.0],dtype=torch.float64,device="cuda:0")
# force the collective path even at world 1
dist.all_reduce(t[:2], op=dist.ReduceOp.MAX); dist.all_reduce(t[2:], op=dist.ReduceOp.SUM); dist.barrier(); torch.cuda.synchronize()
print("nccl world-1 ok", t.tolist(), sharding.summary_dict(t))
dist.destroy_process_group()
