"""The N>1 path of bench.py on ONE GPU: two ranks under torch.distributed.run share cuda:0 (--rehearse-gloo: the framebuffer
reduce goes through gloo on host copies, RCCL wants a device per rank).  Sharding, frames-in-flight per shard, the reduce and
the single JSON line are exercised with the real kernels; the numbers are not bench results."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu():
    common = ["--steps", "1", "--warmup", "0", "--cpu-seconds", "0", "--pmc", "off", "--extra-configs", "off", "--width", "320", "--height", "180", "--spp", "4"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    one = subprocess.run([sys.executable, "bench.py"] + common, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    a = json.loads(one.stdout.strip().splitlines()[-1])
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29533",
                          "bench.py", "--gpus", "2", "--rehearse-gloo"] + common, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert two.returncode == 0, two.stderr[-2000:]
    lines = [l for l in two.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1  # rank 0 alone prints
    b = json.loads(lines[0])
    assert b["n_gpus"] == 2 and b["scaling"] == "weak" and b["unit"] == a["unit"] and b["metric"] == a["metric"]
    # weak scaling: 2x the frames over the same pixels; frames 1..8 instead of 1..4, so about twice the rays
    assert 1.8 < b["config"]["rays_per_step"] / a["config"]["rays_per_step"] < 2.2
    assert "roofline" in b and "pixel tiles x2" in b["config"]["parallelism"]
    # strong scaling: the same 4 frames split by pixel tiles, so exactly the rays of the one-rank run
    three = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29534",
                            "bench.py", "--gpus", "2", "--rehearse-gloo", "--scaling", "strong"] + common, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert three.returncode == 0, three.stderr[-2000:]
    c = json.loads([l for l in three.stdout.strip().splitlines() if l.startswith("{")][0])
    assert c["scaling"] == "strong" and c["config"]["rays_per_step"] == a["config"]["rays_per_step"]
