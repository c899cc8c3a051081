#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X path tracer.

    python bench.py --gpus N --steps K --warmup W [--workload c2|c3|c4|c5] [--scaling strong|weak]

Metric (BASELINE.json): Msamples/s (pixel-samples = camera paths per second), with the roofline
fraction of the dominant kernel.  One *step* = one pass of the hot path over one frame: clear, ONE
fused trace call over every pixel-sample of the frame (pt_prefix + pt_samples_q / pt_samples_w),
resolve to the gamma image and — for N > 1 — the exchange of the tile-sharded radiance buffer to
rank 0 over RCCL.  Scene, camera block and random table are resident in HBM before the timed region.

Workload per N (BASELINE.json `configs`):
  N = 1        C2, the configuration the metric is quoted on: 8 spheres + plane, 1920x1080, 64 spp.
  N = 2, 4, 8  C4 by default — 100 000 spheres + plane, 1920x1080, 64 spp, "tile-sharded 2/4/8 x MI355X":
               STRONG scaling, the same frame cut into 8x8 tiles interleaved over the ranks
               (`--workload c5` is BASELINE's 8-GPU configuration).  The line also carries the same
               workload's single-GPU step time, measured by rank 0 in the same run outside the timed
               region, so that the speed-up can be read from one line.
  --scaling weak   the round-1 form: C2 with 64 x N samples per pixel, per-GPU work fixed — issued as N
               calls of 64 samples, so every rank runs the very kernels of the N = 1 line.

Prints ONE JSON line on rank 0 (contract in the task statement) with the extra objects
  "roofline"     VALU issue: the kernels are bound by vector-instruction issue, not by HBM.
                 achieved = issue cycles of the dominant kernel per launch (rocprofv3 SQ_INSTS_VALU_* class
                 counts x the issue cost of each class measured on this chip, profiles/valu_mix.json)
                 / that kernel's HIP-event duration measured live here; peak = 1024 SIMDs x 2.4 GHz.
                 hbm_frac (counter bytes / time / 8 TB/s) and the round-1 algorithmic-bytes figure
                 (labelled non-physical) ride along.
  "parity"       untimed: probes of the measured configuration AND a crop of the resolved timed frame
                 against the oracle.
  "cpu_baseline" the oracle (CPU restatement of the reference kernel) on this host's cores, bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
SIMDS = 1024               # 256 CUs x 4 SIMD-32
PEAK_CLOCK_GHZ = 2.4
WORKLOAD_DESC = {
    "c2": "C2 Cornell-style: 8 spheres + 1 plane",
    "c3": "C3 textured 12-triangle cube + 4 spheres",
    "c4": "C4 100k random spheres + plane",
    "c5": "C5 50k-triangle dielectric mesh",
}
# dominant kernel of one fused trace call, per workload (what `roofline` prices), and its first stage
KERNELS = {"c2": ("pt_samples_q<false, false, 0", "pt_prefix<false, false>"),
           "c3": ("pt_samples_q<false, false, 1", "pt_prefix<false, false>"),
           "c4": ("pt_samples_q<false, true, 0", "pt_prefix<false, true>"),
           "c5": ("pt_samples_w<false>", "pt_prefix<false, true>")}
CROP = {"c2": (900, 330, 32, 16), "c3": (930, 300, 32, 16), "c4": (960, 270, 8, 4), "c5": (1900, 900, 4, 2)}


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


def source_digest():
    """Digest of the device sources (comments and white space ignored): profiles/valu_mix.json records the one it
    was measured on."""
    import __graft_entry__ as g
    return g.device_source_digest()


def cpu_baseline(wl, table, budget_s=20.0):
    """Time the CPU path on a bounded sample of the same workload: whole rows of the
    frame, all `spp` samples per pixel, as many rows as ~budget_s of CPU work allows
    (calibrated on a thin slice), spread over the frame.  The checker is the oracle port
    (oracle/pt_oracle.c); where the compiled reference kernel (oracle/_ref) is present —
    the build container — that is timed instead."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
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
        """n bands of `rows` rows x cw columns, spread evenly from top to bottom (sky, horizon, floor)."""
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


def roofline(workload, main_ms, first_ms, call_ms, alg_bytes, share=1.0, valid=True):
    """VALU-issue roofline of the dominant kernel from profiles/valu_mix.json (class counts per launch x
    measured issue costs) and the kernel's live HIP-event time; see tools/summarize_profile.py."""
    peak = SIMDS * PEAK_CLOCK_GHZ                                   # G issue cycles / s
    out = {"bound": "valu", "achieved": None, "peak": round(peak, 1),
           "unit": "G VALU issue cycles/s (1024 SIMD-32 x 2.4 GHz; a wave64 instruction occupies its SIMD for 2 "
                   "(fp32 add/mul/fma, int add, logic), 4 (fp64, int mul, shifts, compares, conversions, min/max, "
                   "division helpers, lane reads) or 8 (rcp/sqrt/rsq) cycles: profiles/r02_valu_microbench.md)",
           "frac": None, "traffic": None, "kernel_ms": round(main_ms, 4), "first_stage_ms": round(first_ms, 4),
           "call_ms": round(call_ms, 4)}
    path = os.path.join(ROOT, "profiles", "valu_mix.json")
    mix = None
    if os.path.isfile(path):
        try:
            mix = json.load(open(path)).get(workload)
        except Exception:
            mix = None
    main_key, first_key = KERNELS[workload]
    kern = None
    if mix:
        for name, k in mix.get("kernels", {}).items():
            if name.startswith(main_key):
                kern, out["kernel"] = k, name
    t = main_ms * 1e-3
    if kern and not valid:
        kern = None
        out["note"] = "workload generator arguments overridden: the profiled instruction mix does not apply"
    if kern:
        # N > 1: this rank traces `share` of the frame's pixel-samples (8x8 tiles interleaved: a uniform sample of
        # the image), so it issues that share of the profiled single-GPU launch's instructions
        kern = dict(kern, **{k: kern[k] * share for k in ("issue_cycles", "issue_cycles_low", "issue_cycles_high", "valu_insts")})
        out["share_of_profiled_launch"] = round(share, 5)
        ach = kern["issue_cycles"] / t / 1e9
        lanes = kern.get("lanes_per_inst")
        out.update({
            "achieved": round(ach, 1), "frac": round(ach / peak, 4),
            "frac_bounds": [round(kern["issue_cycles_low"] / t / 1e9 / peak, 4),
                            round(min(kern["issue_cycles_high"] / t / 1e9 / peak, 1.0), 4)],
            "lanes": round(lanes / 64.0, 4) if lanes else None,
            "useful_lane_frac": round(ach / peak * lanes / 64.0, 4) if lanes else None,
            "issue_cycles_per_launch": int(kern["issue_cycles"]), "valu_insts_per_launch": int(kern["valu_insts"]),
            "cycles_per_valu_inst_per_simd": round(t * PEAK_CLOCK_GHZ * 1e9 * SIMDS / kern["valu_insts"], 3),
            "profile": mix.get("profile"), "profile_kernel_ms": kern.get("kernel_ms_profiled"),
            "profile_matches_source": mix.get("source_digest") == source_digest(),
            "scratch_bytes": kern.get("scratch_bytes"), "vgpr": kern.get("vgpr"),
        })
        hbm = sum(k["hbm_bytes"] or 0 for k in mix["kernels"].values()) * share
        if hbm:
            out["traffic"] = int(hbm)
            out["hbm_frac"] = round(hbm / (call_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
    elif "note" not in out:
        out["note"] = "profiles/valu_mix.json has no entry for this workload: run tools/profile.sh + tools/summarize_profile.py"
    # round 1's figure, kept for continuity: the REFERENCE kernel's logical traffic (every primitive struct it
    # would have dereferenced) over the call time.  The scene lives in SGPRs / L2, so these bytes never move:
    # not a physical rate, and not bounded by 1.
    out["alg_hbm_frac"] = {"value": round(alg_bytes / (call_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                           "algorithmic_bytes_per_launch": int(alg_bytes),
                           "note": "non-physical: reference's brute-force logical bytes / time / 8 TB/s (SURVEY 8d)"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOAD_DESC),
                    help="default: c2 at N = 1, c4 (BASELINE's tile-sharded configuration) at N > 1")
    ap.add_argument("--scaling", default="auto", choices=["auto", "strong", "weak"],
                    help="N > 1: strong (default) = the same frame sharded; weak = C2 with 64 x N spp in N calls")
    ap.add_argument("--spp", type=int, default=0, help="override samples per pixel")
    ap.add_argument("--workload-arg", action="append", default=[], metavar="KEY=INT",
                    help="rehearsals only: override a generator argument of the workload, e.g. n_spheres=20000")
    ap.add_argument("--exchange", default="gather", choices=["gather", "reduce"],
                    help="N>1: gather of packed tiles (default) or full-frame RCCL reduce")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity-check", action="store_true")
    ap.add_argument("--no-single-gpu-reference", action="store_true")
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

    scaling = args.scaling if args.scaling != "auto" else "strong"
    workload = args.workload or ("c2" if world == 1 or scaling == "weak" else "c4")
    wl_args = {k: int(v) for k, v in (a.split("=", 1) for a in args.workload_arg)}
    wl = rt.workloads.get(workload, **wl_args)
    base_spp = args.spp or wl.spp
    spp = base_spp * world if scaling == "weak" else base_spp
    chunk = base_spp                                 # samples per fused call: the kernels of the N = 1 line
    tracer = rt.RayTracer(wl.width, wl.height, scene=wl.scene, device=device, seed=rt.workloads.SEED)
    shard = dist_mod.GpuShard(tracer, rank, world, chunk=chunk)
    renderer = dist_mod.ShardedRenderer(shard, rank, world, exchange=args.exchange)
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
    note("counting pass (%s, %d rank(s), %s scaling)" % (workload, world, scaling))
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
    # per-launch kernel durations over the timed region: the library records HIP events on the launch stream
    # around every fused call and between its two stages; they are only read back here
    calls = min(args.steps * (spp // chunk), 64)
    ev_ms = tracer.kernelMsHistory(calls)
    first_ms, main_ms = tracer.stageMsHistory(calls)
    call_ms = float(np.mean(ev_ms))

    t = torch.tensor([wall], dtype=torch.float64, device="cuda")
    s = torch.tensor([float(my_samples)], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
    wall_max, total_samples = float(t.item()), float(s.item())

    # strong scaling: the same workload on ONE GPU, by rank 0, outside the timed region
    single = None
    if world > 1 and scaling == "strong" and not args.no_single_gpu_reference:
        if rank == 0:
            note("single-GPU reference of the same workload (untimed leg)")
            solo = rt.RayTracer(wl.width, wl.height, scene=wl.scene, device=device, seed=rt.workloads.SEED)
            n_solo = 3 if workload in ("c4", "c5") else 10
            solo.renderFrameOnDevice(wl.camera, spp)
            solo.sync()
            t1 = time.perf_counter()
            for _ in range(n_solo):
                solo.renderFrameOnDevice(wl.camera, spp)
            solo.sync()
            single = (time.perf_counter() - t1) / n_solo * 1e3
            solo.close()
        dist.barrier()

    if rank == 0:
        ms_per_step = wall_max / args.steps * 1e3
        value = total_samples * args.steps / wall_max / 1e6
        out = {
            "metric": "Msamples/sec (rays/sec) at %dx%d, %d spp; fraction of the roofline that binds (VALU issue)" %
                      (wl.width, wl.height, spp),
            "value": round(value, 2), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s, %dx%d, %d spp" % (WORKLOAD_DESC[workload], wl.width, wl.height, spp),
                       "width": wl.width, "height": wl.height, "spp_per_call": chunk, "spp_total": spp,
                       "sharding": "8x8 tiles interleaved over %d rank(s)" % world +
                                   ("; %s of the packed radiance tiles to rank 0 over RCCL" % args.exchange if world > 1 else ""),
                       "pixel_samples_per_step": int(total_samples), "seed": hex(rt.workloads.SEED),
                       **({"workload_overrides": wl_args} if wl_args else {})},
            "roofline": roofline(workload, float(np.mean(main_ms)), float(np.mean(first_ms)), call_ms, alg_bytes,
                                 share=my_samples / float(wl.width * wl.height * base_spp) / (spp // chunk),
                                 valid=not wl_args and not args.spp),
            "kernel_ms_min_max": [round(min(ev_ms), 4), round(max(ev_ms), 4)],
            "bounces_per_sample": round(cn.bounces / max(cn.samples, 1), 3),
        }
        if single is not None:
            out["single_gpu_same_workload"] = {"ms_per_step": round(single, 4),
                                               "value": round(total_samples / (single * 1e-3) / 1e6, 2),
                                               "speedup": round(single / ms_per_step, 3),
                                               "note": "rank 0 alone, same frame, same run, outside the timed region"}
        if world == 1 and not args.no_parity_check:
            # untimed sanity leg: the configuration just measured computes the reference's radiance.
            # (1) per pixel-sample probes, bit for bit; (2) a crop of the RESOLVED TIMED FRAME (what pt_prefix +
            # the sample kernel left in the image buffer) against the oracle's progressive image, 1e-4 relative.
            note("parity: probes + crop of the timed frame")
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            from oracle import Oracle
            orc = Oracle()
            rng = np.random.RandomState(1)
            n = 64 if workload in ("c4", "c5") else 2000
            xs, ys, ss = rng.randint(0, wl.width, n), rng.randint(0, wl.height, n), rng.randint(0, spp, n)
            got = tracer.traceSamples(wl.camera, xs, ys, ss)
            exp, _ = orc.samples(wl.scene, wl.camera, table, wl.width, wl.height, xs, ys, ss)
            same = int((got.view(np.uint32) == exp.view(np.uint32)).all(axis=1).sum())
            x0, y0, cw, ch = CROP[workload]
            frame = renderer.image()                  # the last timed step's image
            ref, _ = orc.render(wl.scene, wl.camera, table, wl.width, wl.height, 2, count=spp, region=(x0, y0, cw, ch),
                                threads=host_cores())
            a, b = frame[y0:y0 + ch, x0:x0 + cw], ref[y0:y0 + ch, x0:x0 + cw]
            dev = float((np.abs(a - b) / np.maximum(np.maximum(np.abs(a), np.abs(b)), 1e-6)).max())
            out["parity"] = {"probes": n, "bit_exact": same, "checker": "oracle/pt_oracle.c",
                             "timed_frame_crop": [x0, y0, cw, ch], "crop_max_rel_dev": dev, "crop_ok": dev <= 1e-4,
                             "crop_lit_fraction": float((b[..., :3].sum(-1) > 0).mean())}
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
