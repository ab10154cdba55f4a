"""bench.py's N > 1 branch (interleaved tiles per rank, batched launches, display pass, gather per batch, assembly on the
root, buffer hand-back between the two render streams) rehearsed with two freshly started rank processes on ONE GPU:
RTAMD_BENCH_REHEARSE=1 puts both ranks on cuda:0 and moves the gather through gloo / host memory.  Not a measurement — the
real N > 1 line comes from the driver's multi-GPU node over RCCL — but the same step(), flush() and batching code, and the
same check: the frames the root assembles equal the frame one GPU renders alone (config.assembled_frame_equals_single_gpu_frame)."""
import json
import os
import socket
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.timeout(600)
@pytest.mark.parametrize("fmt,steps,gather_batch,launcher", [("rgba8", 6, 4, "torchrun"), ("f32", 6, 4, "self"), ("rgba8", 3, 1, "torchrun")])
def test_two_rank_rehearsal_assembles_the_single_gpu_frame(fmt, steps, gather_batch, launcher):
    """launcher "torchrun": the driver's command line.  "self": plain `python bench.py --gpus 2`, which starts its own ranks."""
    env = dict(os.environ, RTAMD_BENCH_REHEARSE="1", OMP_NUM_THREADS="4")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    head = ([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
             "--master-port", str(_free_port())] if launcher == "torchrun" else [sys.executable])
    cmd = head + [os.path.join(REPO, "bench.py"), "--gpus", "2", "--workload", "atrium_small",
                  "--steps", str(steps), "--warmup", "3", "--no-cpu-baseline", "--gather-format", fmt, "--gather-batch", str(gather_batch)]
    p = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=500)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]          # rank 0 prints ONE JSON line
    out = json.loads(lines[0])
    cfg = out["config"]
    assert out["n_gpus"] == 2 and cfg["parallelism"] == "tiles2" and cfg["gather_format"] == fmt and cfg["frames_per_launch"] == gather_batch
    assert steps % gather_batch != 0 or gather_batch == 1    # 6 steps in batches of 4: the last batch is partial
    assert cfg["assembled_frame_equals_single_gpu_frame"] is True
    assert "REHEARSAL_not_a_measurement" in cfg
    # the collective and the root's assembly are reported apart from the rendering; gloo here, so no RCCL ranks
    assert cfg["collective_backend"] == "gloo" and cfg["rccl_ranks"] == 0
    assert cfg["gather_ms_per_frame"] > 0 and cfg["assemble_ms_per_frame"] > 0
    assert out["value"] > 0 and out["steps"] == steps and out["scaling"] == "strong"
    # rays are counted over both ranks' shares: the whole frame's primaries
    assert cfg["primary_rays"] == 480 * 270


_RCCL_CHILD = r'''
import sys
import torch
import torch.distributed as dist
sys.path.insert(0, sys.argv[1])
import bench

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev)          # as bench.py does for N > 1
stream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(stream)
g = torch.Generator(device="cpu").manual_seed(5)
for shape, dtype in (((40960,), torch.int32), ((40960, 4), torch.float32)):   # the rgba8 and the f32 gather formats
    send = (torch.randint(-2**31, 2**31 - 1, shape, generator=g, dtype=torch.int64).to(torch.int32) if dtype == torch.int32
            else torch.rand(shape, generator=g, dtype=torch.float32)).to(dev)
    gathered = torch.zeros((1,) + shape, dtype=dtype, device=dev)
    stream.wait_stream(torch.cuda.default_stream(dev))
    for _ in range(3):
        bench.gather_to_root(send, gathered, 0, 1)
    stream.synchronize()
    assert torch.equal(gathered[0].view(torch.int32), send.view(torch.int32)), dtype
stats = torch.tensor([1, 2, 3, 4], dtype=torch.int64, device=dev)
dist.all_reduce(stats)
tt = torch.tensor([1.5, 2.5], dtype=torch.float64, device=dev)
dist.all_reduce(tt, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
assert stats.tolist() == [1, 2, 3, 4] and tt.tolist() == [1.5, 2.5]
dist.destroy_process_group()
print("RCCL_OK", torch.cuda.nccl.version())
'''


@pytest.mark.timeout(300)
def test_rccl_calls_of_the_multi_gpu_path_on_one_rank():
    """The collective calls bench.py makes with N > 1 — init_process_group("nccl", device_id=...), its gather_to_root()
    into row views of the root's buffer on a non-default stream, all_reduce (sum of int64, max of float64), barrier,
    destroy — through RCCL itself with a world of ONE rank: all a one-GPU box can show of the real backend (the two-rank
    rehearsal above runs the batching logic over gloo).  A fresh process, so that no process group outlives the test."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, "-c", _RCCL_CHILD, REPO], cwd=REPO, env=env, capture_output=True, text=True, timeout=280)
    assert p.returncode == 0 and "RCCL_OK" in p.stdout, (p.stdout[-1500:], p.stderr[-3000:])


@pytest.mark.timeout(600)
def test_c5_bench_two_rank_rehearsal_traces_the_single_gpu_ray_count():
    """tools/bench_c5.py --gpus 2 (BASELINE config 5, tile-split) as two fresh rank processes on one GPU over gloo: the ranks'
    shards together trace exactly the rays the single-process run traces (the chain's buffers are compared bit for bit in
    tests/test_gpu_multibounce.py::test_tile_sharded_chain_equals_the_whole_frame_chain)."""
    common = ["--spp", "3", "--width", "496", "--height", "270", "--lanes", "2"]
    one = subprocess.run([sys.executable, os.path.join(REPO, "tools", "bench_c5.py"), *common], cwd=REPO, capture_output=True, text=True, timeout=280)
    assert one.returncode == 0, one.stderr[-2000:]
    a = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][0])
    env = dict(os.environ, RTAMD_BENCH_REHEARSE="1", OMP_NUM_THREADS="4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(REPO, "tools", "bench_c5.py"), "--gpus", "2", *common]
    two = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=280)
    assert two.returncode == 0, two.stderr[-3000:]
    lines = [ln for ln in two.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    b = json.loads(lines[0])
    assert b["n_gpus"] == 2 and b["config"]["parallelism"] == "tiles2" and "REHEARSAL_not_a_measurement" in b
    assert b["config"]["rays"] == a["config"]["rays"] > 3 * 496 * 270 * 2
