#!/usr/bin/env python3
"""one-rank RCCL sanity probe: init_process_group('nccl'), all_reduce / all_gather on device tensors"""
import os, time, sys
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29533')
import torch
import torch.distributed as dist
t0 = time.time()
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
print('init', round(time.time()-t0, 2), flush=True)
x = torch.ones(1000, dtype=torch.float64, device='cuda')
dist.all_reduce(x)
torch.cuda.synchronize()
print('all_reduce', round(time.time()-t0, 2), float(x.sum()), flush=True)
g = [torch.zeros(2, dtype=torch.float64, device='cuda')]
dist.all_gather(g, torch.tensor([1., 2.], dtype=torch.float64, device='cuda'))
torch.cuda.synchronize()
print('all_gather', round(time.time()-t0, 2), g[0].tolist(), flush=True)
for _ in range(200):
    dist.all_reduce(x)
torch.cuda.synchronize()
print('200 all_reduce', round(time.time()-t0, 2), flush=True)
dist.barrier()
print('barrier', round(time.time()-t0, 2), flush=True)
dist.destroy_process_group()
print('done', flush=True)
