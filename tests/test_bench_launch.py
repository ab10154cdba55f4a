"""bench.py's launch contract on a box without the GPUs it is asked for: `--gpus N` must never fall through to fewer GPUs."""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "RTAMD_BENCH_REHEARSE")}
    env.update(kw)
    return env


def test_gpus_n_without_launcher_refuses_a_box_with_fewer_gpus():
    import torch
    want = torch.cuda.device_count() + 2       # more than this box has (the build container has none)
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", str(want), "--steps", "2", "--warmup", "1"],
                       cwd=REPO, env=_env(), capture_output=True, text=True, timeout=120)
    assert p.returncode == 2, (p.returncode, p.stderr[-1000:])
    assert f"--gpus {want}" in p.stderr and "not falling back" in p.stderr
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]      # no JSON line for a run that did not happen


def test_gpus_n_must_agree_with_the_launchers_world_size():
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "4", "--steps", "2", "--warmup", "1"],
                       cwd=REPO, env=_env(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE=1" in p.stderr
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
