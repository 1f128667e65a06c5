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
    assert "roofline" in b and "pixel tiles x2" in b["config"]["parallelism"] and "reduce of the full accumulation buffers" in b["config"]["parallelism"]
    assert "weak scaling: 4 per GPU x 2 GPUs" in b["config"]["workload"]
    # the same line carries the other mode's figure (fixed total spp) and what the collective cost
    o = b["config"]["other_scaling"]
    assert o["scaling"] == "strong" and o["spp_total"] == 4 and o["value"] > 0 and b["config"]["collective_ms_per_step"] >= 0
    assert b["config"]["collective_bytes_into_root"] == 320 * 180 * 16
    # strong scaling: the same 4 frames split by pixel tiles, so exactly the rays of the one-rank run
    three = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29534",
                            "bench.py", "--gpus", "2", "--rehearse-gloo", "--scaling", "strong"] + common, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert three.returncode == 0, three.stderr[-2000:]
    c = json.loads([l for l in three.stdout.strip().splitlines() if l.startswith("{")][0])
    assert c["scaling"] == "strong" and c["config"]["rays_per_step"] == a["config"]["rays_per_step"] and "strong scaling" in c["config"]["workload"]
    # the tile gather instead of the full-buffer reduce: the same rays, 1/N of the bytes into the root
    four = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29535",
                           "bench.py", "--gpus", "2", "--rehearse-gloo", "--scaling", "strong", "--collective", "gather"] + common, cwd=ROOT, env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, text=True, timeout=600)
    assert four.returncode == 0, four.stderr[-2000:]
    g = json.loads([l for l in four.stdout.strip().splitlines() if l.startswith("{")][0])
    assert g["config"]["rays_per_step"] == a["config"]["rays_per_step"] and "gather of every rank's own tiles" in g["config"]["parallelism"]
    assert g["config"]["collective_bytes_into_root"] == 320 * 180 * 16 // 2


@pytest.mark.gpu
def test_bench_line_contract_with_its_own_pmc_passes():
    """The driver's line: contract keys, a roofline for the dominant kernel whose fractions come from rocprofv3 passes made by the run
    itself (frac <= 1 by construction, in the unit of the bound it names), per-kernel table, CPU baselines incl. the single thread."""
    r = subprocess.run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--width", "640", "--height", "360", "--spp", "8", "--cpu-seconds", "1",
                        "--extra-configs", "off"], cwd=ROOT, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "Mrays/s" and d["n_gpus"] == 1 and d["steps"] == 2 and d["dtype"] == "f32" and d["value"] > 0
    roof = d["roofline"]
    assert roof["kernel"] in ("k_generate", "k_tail", "k_bvh", "k_shade", "k_accumulate") and roof["avg_launch_ms"] > 0
    tab = roof["kernels"]
    assert abs(sum(v["share_of_kernel_time"] for v in tab.values()) - 1.0) < 1e-6
    assert roof["kernel"] == max(tab, key=lambda k: tab[k]["ms_per_step"])
    if "pmc_note" not in roof:  # rocprofv3 is on the GPU box: the passes must have produced the fractions
        # fractions are NOT clamped any more: a model that passes 1 must show it; on this small batch they stay well below 1.25
        assert roof["bound"] in ("valu_issue", "hbm", "l1_gather", "vector_memory_path") and 0 < roof["frac"] < 1.25
        assert roof["traffic"] > 0 and "rocprofv3 --pmc passes made by this run" in roof["pmc_source"]
        assert "valu_busy_frac_at_2p4_ghz" in tab[roof["kernel"]]
        for k in ("step_fabric_bytes", "compulsory_bytes", "state_traffic_bytes"):
            assert roof[k] == roof[k] and roof[k] is not None, k
        assert roof["compulsory_bytes"] >= 32 * 640 * 360 * 8
        for v in tab.values():
            if "valu_busy_frac_at_2p4_ghz" in v:  # (a batch this small is k_generate + k_tail + k_accumulate: the per-bounce kernels are never launched)
                assert 0 <= v["valu_busy_frac_at_2p4_ghz"] < 1.25 and 0 <= v["fabric_frac_of_hbm_peak"] < 1.25 and 0 < v["active_lane_frac"] <= 1
                assert v["rocprof_valu_busy"] > 0 and v["valu_busy_frac_at_pass_clock"] > 0
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and cb["single_thread"]["cores"] == 1 and cb["single_thread"]["value"] > 0


@pytest.mark.gpu
def test_bench_in_library_multi_device_path_without_a_launcher():
    """`python bench.py --gpus N` with no torch.distributed.run around it: ONE process, ONE context over the N GPUs (ptmi_create_multi), the reduce
    inside the library (ptmi_reduce_framebuffer).  Rehearsed on one GPU with --devices 0,0 (the shards share the GPU and are summed by a kernel)."""
    common = ["--steps", "1", "--warmup", "0", "--cpu-seconds", "0", "--pmc", "off", "--extra-configs", "off", "--width", "320", "--height", "180", "--spp", "4"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    one = subprocess.run([sys.executable, "bench.py"] + common, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    a = json.loads(one.stdout.strip().splitlines()[-1])
    two = subprocess.run([sys.executable, "bench.py", "--devices", "0,0", "--scaling", "strong"] + common, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert two.returncode == 0, two.stderr[-2000:]
    lines = [l for l in two.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    b = json.loads(lines[0])
    assert b["n_gpus"] == 2 and b["scaling"] == "strong" and b["metric"] == a["metric"] and b["value"] > 0
    assert "ptmi_create_multi x2" in b["config"]["parallelism"] and "tile gather" in b["config"]["parallelism"]
    assert b["config"]["other_scaling"]["scaling"] == "weak"
    assert b["config"]["rays_per_step"] == a["config"]["rays_per_step"]  # the same 4 frames, pixel tiles dealt to the two shards
    # weak scaling (the default): spp x 2
    w = subprocess.run([sys.executable, "bench.py", "--devices", "0,0"] + common, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert w.returncode == 0, w.stderr[-2000:]
    c = json.loads([l for l in w.stdout.strip().splitlines() if l.startswith("{")][0])
    assert c["scaling"] == "weak" and 1.8 < c["config"]["rays_per_step"] / a["config"]["rays_per_step"] < 2.2
    # asking for more GPUs than the box has fails loudly (the driver must see it), it does not fall back to one
    bad = subprocess.run([sys.executable, "bench.py", "--gpus", "64"] + common, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert bad.returncode != 0


@pytest.mark.gpu
@pytest.mark.parametrize("how", ["hang", "crash"])
def test_bench_watchdog_falls_back_when_the_first_attempt_hangs_or_dies(how):
    """The first contact with RCCL on an 8-GPU node must not be able to lose the record: for N > 1 bench.py measures in a fresh child
    process; when that child hangs (PTMI_BENCH_SIMULATE=hang: it sleeps forever) or dies (=crash), the parent — which never touched the GPU — kills it by its
    process group and starts ONE more fresh child with the fall-back collective; the line it prints says FALLBACK and why.  In-library driver on one GPU
    (--devices 0,0) and the one-process-per-GPU driver (two ranks under torch.distributed.run, gloo rehearsal)."""
    common = ["--steps", "1", "--warmup", "0", "--cpu-seconds", "0", "--pmc", "off", "--extra-configs", "off", "--width", "320", "--height", "180", "--spp", "4", "--watchdog-seconds", "60"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PTMI_BENCH_SIMULATE=how)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "bench.py", "--devices", "0,0"] + common, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    par = d["config"]["parallelism"]
    assert d["n_gpus"] == 2 and d["value"] > 0 and "FALLBACK after the first attempt failed" in par, par
    assert ("no line within 60 s" in par) if how == "hang" else ("exited with code 3" in par), par
    assert "watchdog" in r.stderr
    if how == "crash":  # (the hang costs a minute per driver: once is enough)
        two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29541",
                              "bench.py", "--gpus", "2", "--rehearse-gloo"] + common, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
        assert two.returncode == 0, two.stderr[-2000:]
        lines = [l for l in two.stdout.strip().splitlines() if l.startswith("{")]
        assert len(lines) == 1
        b = json.loads(lines[0])
        par = b["config"]["parallelism"]
        assert b["n_gpus"] == 2 and "FALLBACK after the first attempt failed" in par and "gather of every rank's own tiles" in par and "gloo" in par, par
