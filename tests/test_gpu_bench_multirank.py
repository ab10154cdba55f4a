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
@pytest.mark.parametrize("fmt,steps,gather_batch", [("rgba8", 6, 4), ("f32", 6, 4), ("rgba8", 3, 1)])
def test_two_rank_rehearsal_assembles_the_single_gpu_frame(fmt, steps, gather_batch):
    env = dict(os.environ, RTAMD_BENCH_REHEARSE="1", OMP_NUM_THREADS="4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(REPO, "bench.py"), "--gpus", "2", "--workload", "atrium_small",
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
    assert out["value"] > 0 and out["steps"] == steps and out["scaling"] == "strong"
    # rays are counted over both ranks' shares: the whole frame's primaries
    assert cfg["primary_rays"] == 480 * 270
