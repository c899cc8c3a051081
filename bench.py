#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X path tracer.

    python bench.py --gpus N --steps K --warmup W [--workload c2|c3|c4|c5]

Metric (BASELINE.json): Msamples/s at 1920×1080, 64 spp, and the fraction of the
HBM roofline.  One *step* = one pass of the hot path over one frame: clear,
ONE fused launch tracing every pixel-sample of the frame (samples 0..spp-1 of
every pixel), resolve to the gamma image, and — for N > 1 — the RCCL reduce of
the tile-sharded radiance buffer to rank 0.  Scene, camera block and random
table are resident in HBM before the timed region starts.

N > 1 (launched by torch.distributed.run, one rank per GPU): the 1920×1080 frame
is cut into 8×8 tiles interleaved over ranks; per-GPU work is held fixed (weak
scaling) by giving every pixel 64·N samples, so each rank traces 132.7 M
pixel-samples per step exactly as at N = 1.  value = pixel-samples all ranks traced
÷ max-over-ranks wall time.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra
objects: "roofline" (algorithmic bytes of the reference kernel per launch ÷ the
kernel's HIP-event duration, against the 8 TB/s HBM peak) and "cpu_baseline" (the
oracle / compiled reference kernel timed on this host's cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
WORKLOAD_DESC = {
    "c2": "C2 Cornell-style: 8 spheres + 1 plane, 1920x1080, 64 spp",
    "c3": "C3 textured 12-triangle cube + 4 spheres, 1920x1080, 256 spp",
    "c4": "C4 100k random spheres + plane, 1920x1080, 64 spp",
    "c5": "C5 50k-triangle dielectric mesh, 3840x2160, 512 spp",
}


def host_cores():
    """CPU share of this process: cgroup quota if one is set (the GPU box gives 16 cores
    of a 256-thread host), else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 64))


def cpu_baseline(wl, table, budget_s=20.0):
    """Time the CPU path on a bounded sample of the same workload: whole rows of the
    frame, all `spp` samples per pixel, as many rows as ~budget_s of CPU work allows
    (calibrated on a thin slice), spread over the frame.  Uses oracle/_ref (the
    compiled reference kernel) when that library is present, else the oracle port."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    from oracle import Oracle, Reference
    threads = host_cores()
    kind = "reference" if Reference.available() else "port"
    ref = Reference() if kind == "reference" else None
    orc = Oracle()

    def run(regions, spp):
        """trace + (spp-1) retrace on each (x0, y0, cw, ch) region."""
        t0 = time.perf_counter()
        for region in regions:
            if ref is not None:
                ref.progressive(wl.scene, wl.camera, table, wl.width, wl.height, spp, threads, region=region)
            else:
                orc.render(wl.scene, wl.camera, table, wl.width, wl.height, 2, count=spp, region=region,
                           threads=threads)
        return time.perf_counter() - t0

    W, H, spp = wl.width, wl.height, wl.spp
    rows = min(threads, H)

    def bands(n, cw):
        """n bands of `rows` rows × cw columns, spread evenly from top to bottom (sky, horizon, floor)."""
        x0 = (W - cw) // 2
        return [(x0, int(i * (H - rows) / max(n - 1, 1)) if n > 1 else (H - rows) // 2, cw, rows) for i in range(n)]

    # calibrate the cost per pixel-sample on a growing probe (a 100k-sphere bounce costs ~1 ms of CPU)
    cw, t_cal = 4, 0.0
    while True:
        t_cal = run(bands(4, cw), 1)
        if t_cal > 0.3 or cw >= W:
            break
        cw = min(W, cw * 4)
    core_s_per_sample = max(t_cal, 1e-4) * threads / (4 * rows * cw)
    n_target = budget_s / core_s_per_sample                      # pixel-samples the budget buys
    max_bands = max(1, H // rows)
    if n_target >= W * rows * spp:                               # whole-width bands at full spp
        nb, cw_s, spp_s = int(min(max_bands, n_target // (W * rows * spp))), W, spp
    elif n_target >= 16 * rows * spp:                            # one narrower band at full spp
        nb, cw_s, spp_s = 1, int(n_target // (rows * spp)), spp
    else:                                                        # a 16-column band at reduced spp
        nb, cw_s, spp_s = 1, 16, int(max(1, n_target // (16 * rows)))
    regions = bands(nb, cw_s)
    wall = run(regions, spp_s)
    samples = nb * rows * cw_s * spp_s
    return {"value": round(samples / wall / 1e6, 4), "unit": "Msamples/s", "cores": threads, "kind": kind,
            "sample": "%d band(s) of %d rows x %d columns of the %dx%d frame, %d of %d spp = %.3f M pixel-samples, "
                      "%.1f s wall" % (nb, rows, cw_s, W, H, spp_s, spp, samples / 1e6, wall)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOAD_DESC))
    ap.add_argument("--spp", type=int, default=0, help="override samples per pixel (per GPU)")
    ap.add_argument("--exchange", default="gather", choices=["gather", "reduce"],
                    help="N>1: gather of packed tiles (default) or full-frame RCCL reduce")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity-check", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=20.0, help="seconds of CPU work for the baseline leg")
    args = ap.parse_args()

    import numpy as np
    import torch
    import opencl_raytracing_amd as rt
    from importlib import import_module
    dist_mod = import_module("opencl-raytracing_amd.distributed")

    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: no HIP device visible (the product has no CPU path)")
    rank, world, local = dist_mod.init_process_group()
    if world != args.gpus:
        sys.exit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run --nproc-per-node %d)" %
                 (args.gpus, world, args.gpus))
    device = local % torch.cuda.device_count()   # == local on a full node; rehearsals may oversubscribe a GPU
    torch.cuda.set_device(device)
    import torch.distributed as dist

    wl = rt.workloads.get(args.workload)
    base_spp = args.spp or wl.spp
    spp = base_spp * world                       # weak scaling: per-GPU pixel-samples fixed
    tracer = rt.RayTracer(wl.width, wl.height, scene=wl.scene, device=device, seed=rt.workloads.SEED)
    renderer = dist_mod.ShardedRenderer(dist_mod.GpuShard(tracer, rank, world), rank, world, exchange=args.exchange)
    table = tracer.getRandomTable() if rank == 0 else None

    def step():
        renderer.render(wl.camera, spp)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def note(msg):
        if rank == 0:
            print("[bench] " + msg, file=sys.stderr, flush=True)

    # algorithmic bytes of one launch: exact work counters from the counting build (untimed)
    note("counting pass")
    tracer.enableCounters(True)
    tracer.resetCounters()
    step()
    tracer.sync()
    cn = tracer.counters()
    tracer.enableCounters(False)
    alg_bytes = cn.algorithmic_bytes()
    my_samples = cn.samples

    note("warmup + %d timed steps" % args.steps)
    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    wall = time.perf_counter() - t0
    # per-launch kernel duration over the timed region: the library records a HIP event
    # pair on the launch stream around every trace launch; read them back afterwards
    ev_ms = tracer.kernelMsHistory(min(args.steps, 64))
    launch_ms = float(np.mean(ev_ms))

    t = torch.tensor([wall], dtype=torch.float64, device="cuda")
    s = torch.tensor([float(my_samples)], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
    wall_max, total_samples = float(t.item()), float(s.item())

    if rank == 0:
        ms_per_step = wall_max / args.steps * 1e3
        value = total_samples * args.steps / wall_max / 1e6
        achieved = alg_bytes / (launch_ms * 1e-3) / 1e9
        flops = 17 * cn.t_sphere + 11 * cn.t_plane + 60 * cn.t_lens + 35 * cn.t_tri + 60 * cn.h_bounce
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.isfile(tpath):
            try:
                traffic = json.load(open(tpath)).get(args.workload, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "Msamples/sec (rays/sec) at 1920x1080, 64 spp; % HBM roofline",
            "value": round(value, 2), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": WORKLOAD_DESC[args.workload], "width": wl.width, "height": wl.height,
                       "spp_per_gpu": base_spp, "spp_total": spp, "sharding": "8x8 tiles interleaved over %d rank(s)"
                       % world + ("; RCCL %s of the radiance buffer to rank 0" % args.exchange if world > 1 else ""),
                       "pixel_samples_per_step": int(total_samples), "seed": hex(rt.workloads.SEED)},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "kernel": "pt_prefix + %s (one fused trace call)" % ("pt_samples_w" if args.workload == "c5" else "pt_samples_q"), "kernel_ms": round(launch_ms, 4),
                         "algorithmic_bytes_per_launch": int(alg_bytes),
                         "bytes_per_pixel_sample": round(alg_bytes / max(my_samples, 1), 1),
                         "bounces_per_sample": round(cn.bounces / max(cn.samples, 1), 3),
                         # SURVEY §8d's side figure: useful fp32 work of the REFERENCE's loops at its per-test
                         # flop counts (sphere 17, plane 11, lens 60, triangle 35, shading 60 per hit) against the
                         # vector peak without FMA (the parity contract forbids contraction): 157.3 / 2 TFLOP/s
                         "valu": {"flop_per_launch_est": int(flops), "achieved": round(flops / (launch_ms * 1e-3) / 1e12, 2),
                                  "peak": 78.6, "unit": "TFLOP/s (fp32 vector, no FMA)",
                                  "frac": round(flops / (launch_ms * 1e-3) / 1e12 / 78.6, 4)}},
            "kernel_ms_min_max": [round(min(ev_ms), 4), round(max(ev_ms), 4)],
        }
        if world == 1 and not args.no_parity_check:
            # untimed sanity leg: the configuration just measured computes the reference's radiance —
            # probes of the frame against the oracle (checker only, never the thing measured)
            note("parity probes")
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            from oracle import Oracle
            rng = np.random.RandomState(1)
            n = 64 if args.workload in ("c4", "c5") else 2000
            xs, ys, ss = rng.randint(0, wl.width, n), rng.randint(0, wl.height, n), rng.randint(0, spp, n)
            got = tracer.traceSamples(wl.camera, xs, ys, ss)
            exp, _ = Oracle().samples(wl.scene, wl.camera, table, wl.width, wl.height, xs, ys, ss)
            same = int((got.view(np.uint32) == exp.view(np.uint32)).all(axis=1).sum())
            out["parity"] = {"probes": n, "bit_exact": same, "checker": "oracle/pt_oracle.c"}
        if not args.no_cpu_baseline and world == 1:
            note("cpu baseline")
            out["cpu_baseline"] = cpu_baseline(wl, table, args.cpu_budget)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    tracer.close()


if __name__ == "__main__":
    main()
