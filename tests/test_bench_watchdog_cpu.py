"""bench.py's watchdog for N > 1, as far as a box without a GPU can take it: the parent never touches the GPU (it must work here), starts the measurement
as a fresh child, and when that child dies starts exactly ONE more with the fall-back settings; when that one fails too (here: there is no GPU) it gives up with a
non-zero exit code instead of hanging or looping.  The GPU half — a line marked FALLBACK — is tests/test_bench_dist_gpu.py."""
import os
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_watchdog_tries_once_more_and_then_gives_up():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the second attempt would succeed (covered by tests/test_bench_dist_gpu.py)")
    env = dict(os.environ, PTMI_BENCH_SIMULATE="crash")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    t = time.time()
    r = subprocess.run([sys.executable, "bench.py", "--devices", "0,0", "--steps", "1", "--warmup", "0", "--cpu-seconds", "0", "--pmc", "off", "--extra-configs", "off",
                        "--width", "64", "--height", "64", "--spp", "1", "--watchdog-seconds", "120"], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=400)
    assert r.returncode == 1, (r.returncode, r.stderr[-1500:])
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]  # no line was invented
    notes = [l for l in r.stderr.splitlines() if l.startswith("bench.py watchdog")]
    assert len(notes) == 2 and "attempt 0 exited with code 3" in notes[0] and "attempt 1 exited with code" in notes[1], r.stderr[-1500:]
    assert time.time() - t < 300


def test_watchdog_kills_a_hung_child_by_its_process_group():
    """PTMI_BENCH_SIMULATE=hang: the first child sleeps forever; the parent must be back after about 2 x the limit (the second attempt dies at once here: no GPU), and the
    sleeping child must be gone."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present (covered by tests/test_bench_dist_gpu.py)")
    env = dict(os.environ, PTMI_BENCH_SIMULATE="hang")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    t = time.time()
    r = subprocess.run([sys.executable, "bench.py", "--devices", "0,0", "--steps", "1", "--warmup", "0", "--cpu-seconds", "0", "--pmc", "off", "--extra-configs", "off",
                        "--width", "64", "--height", "64", "--spp", "1", "--watchdog-seconds", "8"], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 1, (r.returncode, r.stderr[-1500:])
    notes = [l for l in r.stderr.splitlines() if l.startswith("bench.py watchdog")]
    assert len(notes) == 2 and "no line within 8 s (killed)" in notes[0], r.stderr[-1500:]
    assert time.time() - t < 120
    left = subprocess.run(["ps", "-eo", "pid,args"], stdout=subprocess.PIPE, text=True).stdout
    assert not [l for l in left.splitlines() if "bench.py" in l and "--worker" in l and "--attempt 0" in l], left
