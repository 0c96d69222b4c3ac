"""world_size-1 RCCL smoke (the only RCCL configuration a 1-GPU box can run): same
process-group options as bench.py, TorchComm.alltoallv to self, sync and async."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import fusgpu_loader  # noqa: E402

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
dist.init_process_group("nccl", device_id=dev, pg_options=opts)
scat = fusgpu_loader.submodule("scatterer")
comm = scat.TorchComm()
assert comm.size == 1 and comm.backend == "nccl"
send = torch.arange(1000, dtype=torch.float64, device=dev)
recv = torch.zeros(1000, dtype=torch.float64, device=dev)
comm.alltoallv(send, [1000], recv, [1000])
assert torch.equal(send, recv)
recv.zero_()
w = comm.alltoallv(send * 2, [1000], recv, [1000], async_op=True)
w.wait()
torch.cuda.synchronize()
assert torch.equal(send * 2, recv)
idx = comm.alltoallv_int64(np.arange(7, dtype=np.int64), np.array([7]), np.array([7]))
assert np.array_equal(idx, np.arange(7))
# empty exchange (a rank with no ghosts on one side)
e = torch.zeros(0, dtype=torch.float64, device=dev)
comm.alltoallv(e, [0], e.clone(), [0])
torch.cuda.synchronize()
dist.barrier()
dist.destroy_process_group()
print("NCCL_WORLD1_OK")
