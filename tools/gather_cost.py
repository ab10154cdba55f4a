#!/usr/bin/env python3
"""Experiment (GPU box, one rank): host cost per call of torch.distributed's RCCL gather, which
rank 0 pays once per frame in bench.py's N > 1 path.  World size 1, so no link traffic: what is
left is the c10d / RCCL call overhead the host adds to every step."""
import os, time
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st)
for n in (260_000, 1_040_000):
    src = torch.zeros((n, 4), dtype=torch.float32, device=dev)
    dst = torch.empty((1, n, 4), dtype=torch.float32, device=dev)
    for name, fn in (("gather", lambda: dist.gather(src, gather_list=list(dst.unbind(0)), dst=0)),
                     ("all_gather_into_tensor", lambda: dist.all_gather_into_tensor(dst.view(-1, 4), src))):
        for _ in range(20): fn()
        torch.cuda.synchronize(); K = 500; t0 = time.perf_counter()
        for _ in range(K): fn()
        t_cpu = time.perf_counter() - t0; torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"{name} {n * 16 / 1e6:.1f} MB: host issue {t_cpu / K * 1e6:.1f} us per call, {dt / K * 1e6:.1f} us per call wall", flush=True)
dist.destroy_process_group()
