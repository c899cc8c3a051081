"""bench.py's contract on the GPU box: the one-GPU line with `roofline`, `parity` and `cpu_baseline`,
and the N > 1 launch rehearsed with two ranks sharing the one GPU (gloo moves the buffers — RCCL
refuses two ranks on one device; the kernels, sharding, packing and timing code are the real ones)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _json_line(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_bench_one_gpu_line():
    p = subprocess.run([sys.executable, "bench.py", "--steps", "3", "--warmup", "1", "--cpu-budget", "2"], cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    j = _json_line(p.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "arith", "hbm_roofline_pct"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 3 and j["warmup"] == 1 and j["unit"] == "Msamples/s"
    assert j["metric"] == "Msamples/sec (rays/sec) at 1920\u00d71080, 64 spp; % HBM roofline"     # BASELINE.json's metric
    assert j["scaling"] == "weak" and j["config"]["workload"].startswith("C2")
    assert j["config"]["pixel_samples_per_step"] == 1920 * 1080 * 64
    assert j["arith"] == "rocm-opencl"           # the timed arithmetic: the reference as ROCm's OpenCL builds it
    r = j["roofline"]
    # the roofline that binds: VALU issue, counter-derived (profiles/valu_mix.json) over the live kernel time — a
    # physical fraction, so it cannot exceed 1; HBM, work / peak and the round-1 algorithmic-bytes figure ride along
    assert r["bound"] == "valu" and r["frac"] is not None and 0.0 < r["frac"] <= 1.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert abs(r["achieved"] - r["issue_cycles_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9) < 0.5
    assert r["frac_bounds"][0] <= r["frac"] <= r["frac_bounds"][1] <= 1.0
    assert 0.0 < r["lanes"] <= 1.0 and 0.0 < r["hbm_frac"] < 1.0
    assert abs(j["hbm_roofline_pct"] - 100 * r["hbm_frac"]) < 1e-2
    assert isinstance(r["profile_matches_source"], bool)   # False = profiles/valu_mix.json predates a kernel edit: re-profile
    assert "non-physical" in r["alg_hbm_frac"]["note"]
    assert 0.05 < r["alg_flop_frac"]["value"] < 1.0
    assert abs(r["kernel_ms"] + r["first_stage_ms"] - r["call_ms"]) < 0.05 * r["call_ms"]
    # the step is the launch plus clear + resolve: the wall clock per step cannot be below the kernel time
    assert j["ms_per_step"] >= r["call_ms"] * 0.98
    assert abs(j["value"] - j["config"]["pixel_samples_per_step"] / (j["ms_per_step"] * 1e-3) / 1e6) < 0.01 * j["value"]
    # parity, untimed: the timed policy against the reference's real OpenCL build, policy 0 against the CPU oracle
    par = j["parity"]
    assert par["arith_timed"] == "rocm-opencl" and par["walk_overflow"] == 0
    ro = par["real_opencl"]
    assert ro["ok"] and ro["pixel_samples_bit_identical_samples_0_1"] == [1.0, 1.0]
    assert ro["timed_frame_max_rel_dev"] <= 1e-4 and ro["lit_fraction"] > 0.2
    co = par["cpu_oracle_ieee"]
    assert co["bit_exact"] == co["probes"] > 0 and co["crop_ok"] and co["crop_max_rel_dev"] <= 1e-4 and co["crop_lit_fraction"] > 0.2
    assert set(j["arith_variants"]) == {"ieee", "rocm-opencl-nocontract", "rocm-opencl"}
    assert all(v["kernel_ms"] > 0 for v in j["arith_variants"].values())
    c = j["cpu_baseline"]
    assert c["value"] > 0 and c["cores"] >= 1 and c["kind"] in ("reference", "port")
    assert j["value"] > 100 * c["value"]


def test_bench_other_workloads_sizes_and_arithmetics():
    """north_star's own line — the MESH scene at 1920x1080 x 64 spp — and the ieee policy: --workload / --size / --spp /
    --arith; each has its own entry in profiles/valu_mix.json."""
    p = subprocess.run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--workload", "c5", "--size", "1920x1080",
                        "--spp", "64", "--no-cpu-baseline", "--no-arith-variants"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    j = _json_line(p.stdout)
    assert j["metric"].startswith("Msamples/sec (rays/sec) at 1920\u00d71080, 64 spp") and j["config"]["workload"].startswith("C5")
    assert j["config"]["profile_key"] == "c5@1920x1080x64" and j["scaling"] == "strong"
    assert j["parity"]["real_opencl"]["ok"] and j["parity"]["cpu_oracle_ieee"]["crop_ok"] and j["parity"]["walk_overflow"] == 0
    p = subprocess.run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--arith", "ieee", "--no-cpu-baseline",
                        "--no-arith-variants"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    j = _json_line(p.stdout)
    assert j["arith"] == "ieee" and j["parity"]["cpu_oracle_ieee"]["crop_ok"]
    assert 0.9 < min(j["parity"]["real_opencl"]["pixel_samples_bit_identical_samples_0_1"]) < 1.0   # distance, not identity


def _two_ranks(extra, port=None, plain=False):
    """port given: under torch.distributed.run as the driver launches N > 1; plain: `python bench.py --gpus 2`, which
    starts its own ranks.  Both ranks share the one GPU (gloo moves the buffers; RCCL refuses two ranks on one device)."""
    env = dict(os.environ, RT_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    tail = ["bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--cpu-budget", "4"] + extra
    if plain:
        cmd = [sys.executable] + tail
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
               "127.0.0.1", "--master-port", str(port)] + tail
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    return _json_line(p.stdout)


def test_bench_plain_gpus_2_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no torchrun environment — the form of the driver's N = 1 command: the ranks are
    started as a child process and its one JSON line comes back.  Default workload at every N: C2, weak scaling (64 x N
    spp per pixel in one fused call per rank)."""
    j = _two_ranks([], plain=True)
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["config"]["workload"].startswith("C2")
    assert j["config"]["spp_total"] == 128 and j["config"]["spp_per_call"] == 128
    assert j["config"]["pixel_samples_per_step"] == 1920 * 1080 * 128
    assert j["single_gpu_same_workload"]["ms_per_step"] > 0 and j["speedup_vs_1gpu_same_workload"] > 0
    assert j["cpu_baseline"]["value"] > 0 and j["walk_overflow"] == 0
    assert j["roofline"]["bound"] == "valu"


def test_bench_plain_gpus_2_propagates_failure():
    """A rank that fails makes the self-launched run fail: the child's return code comes back, no JSON line is invented."""
    env = dict(os.environ, RT_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0", "--size", "0x0"], cwd=ROOT,
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode != 0
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]


@pytest.mark.parametrize("exchange", ["gather", "reduce"])
def test_bench_two_ranks_strong_scaling_of_c4(exchange):
    """--workload c4: BASELINE's tile-sharded configuration (here with 20 000 of its 100 000 spheres to keep the
    rehearsal short), the SAME frame cut over the ranks, launched as the driver launches N > 1."""
    j = _two_ranks(["--workload", "c4", "--exchange", exchange, "--workload-arg", "n_spheres=20000", "--no-cpu-baseline"], port=29533)
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["config"]["workload"].startswith("C4")
    assert j["config"]["spp_total"] == 64 and j["config"]["pixel_samples_per_step"] == 1920 * 1080 * 64
    assert j["value"] > 0 and j["single_gpu_same_workload"]["ms_per_step"] > 0 and j["speedup_vs_1gpu_same_workload"] > 0
    assert j["roofline"]["bound"] == "valu"


def test_sharded_frame_equals_unsharded_frame():
    """Three ranks on the one GPU (gloo): tile-sharded render + exchange == the unsharded render."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr",
           "127.0.0.1", "--master-port", "29541", os.path.join(ROOT, "tests", "dist_gpu_worker.py")]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    assert "SHARDED_OK gather" in p.stdout and "SHARDED_OK reduce" in p.stdout


def test_rccl_accepts_the_collectives_of_the_exchange():
    """World size 1 over RCCL (backend "nccl"): the gather / reduce / barrier calls ShardedRenderer issues,
    on a non-default stream, are accepted by this image's torch + RCCL."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "nccl_selftest.py")], cwd=ROOT, capture_output=True,
                       text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-3000:]
    assert "RCCL world-1 gather/reduce/barrier OK nccl" in p.stdout
