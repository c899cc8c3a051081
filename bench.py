#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X path tracer.

    python bench.py --gpus N --steps K --warmup W
                    [--workload c2|c3|c4|c5] [--size WxH] [--spp S] [--scaling weak|strong]
                    [--arith rocm-opencl|rocm-opencl-nocontract|ieee]

Metric (BASELINE.json): Msamples/s (pixel-samples = camera paths per second) and the % of the HBM roofline; the
roofline that actually binds (VALU issue) rides along.  One *step* = one pass of the hot path over one frame: clear,
the fused trace call(s) over every pixel-sample of the frame (pt_prefix + pt_samples_q / pt_samples_w), resolve to the
gamma image and — for N > 1 — the exchange of the tile-sharded radiance buffer to rank 0 over RCCL.  Scene, camera
block and random table are resident in HBM before the timed region.

Workload: ONE workload along the whole N curve — C2 by default, the configuration the metric is quoted on
(8 spheres + plane, 1920x1080, 64 spp).
  C2 / C3 (frames of milliseconds)  WEAK scaling: every pixel gets 64 x N samples in ONE fused call per rank, the
            frame's 8x8 tiles interleaved over the ranks — each GPU traces the pixel-samples of one N = 1 frame (an
            N-th of the pixels, N times the samples) with the kernels of the N = 1 line, then the packed tiles meet on
            rank 0.  (One call, not N calls of 64: a launch over an N-th of the pixels is mostly tail — measured on one
            GPU, tools/shard_chunk_probe.py: the rank-0 share of an 8-rank step takes 3.44 ms as 8 calls of 64 samples,
            1.42 ms as one call of 512.)
  C4 / C5 (BASELINE's tile-sharded configurations, frames of 0.1 ... 1 s)  STRONG scaling: the same frame cut over the ranks.
Every N > 1 line also carries the same workload's single-GPU step (rank 0 alone, same run, untimed leg) and
`speedup_vs_1gpu_same_workload`, so the curve can be read from one line.
A plain `python bench.py --gpus N` (N > 1, no torchrun environment) starts its own ranks:
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same arguments>` as a CHILD process whose
JSON line and return code are relayed.

`--arith` selects the arithmetic policy that is timed (csrc/pt_arith.hpp): `rocm-opencl` (default) computes, bit for
bit per pixel-sample, what the reference's kernel file computes when ROCm's own OpenCL tool chain builds it with default
options for this chip; `ieee` is the plain-IEEE contract the CPU oracle restates.

Prints ONE JSON line on rank 0 (contract in the task statement) with the extra objects
  "roofline"     bound "valu": the kernels are bound by vector-instruction issue, not by HBM.  achieved =
                 issue cycles of the dominant kernel per launch (rocprofv3 SQ_INSTS_VALU_* class counts x the issue
                 cost of each class measured on this chip, profiles/valu_mix.json) / that kernel's HIP-event duration
                 measured live here; peak = 1024 SIMDs x 2.4 GHz.  hbm_frac (counter bytes / time / 8 TB/s = the
                 metric's "% HBM roofline"), alg_flop_frac (the reference's logical flops / time / 157.3 TFLOP/s) and the
                 round-1 algorithmic-bytes figure (labelled non-physical) ride along.
  "parity"       untimed: `real_opencl` — the timed policy against the reference's gfx950 OpenCL code object, per
                 pixel-sample bit for bit and the timed frame within 1e-4; `cpu_oracle_ieee` — policy 0 against the
                 CPU oracle (probes bit for bit, a frame crop within 1e-4); `walk_overflow` must be 0.
  "cpu_baseline" the oracle (CPU restatement of the reference kernel) on this host's cores, bounded sample.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
FP32_PEAK_TFLOPS = 157.3   # vector fp32 (not MFMA): 256 CUs x 128 lanes x 2 flop x 2.4 GHz
SIMDS = 1024               # 256 CUs x 4 SIMD-32
PEAK_CLOCK_GHZ = 2.4
WORKLOAD_DESC = {
    "c2": "C2 Cornell-style: 8 spheres + 1 plane",
    "c3": "C3 textured 12-triangle cube + 4 spheres",
    "c4": "C4 100k random spheres + plane",
    "c5": "C5 50k-triangle dielectric OBJ mesh",
}
ARITH = {"ieee": 0, "rocm-opencl-nocontract": 1, "rocm-opencl": 2}
# dominant kernel of one fused trace call, per workload (what `roofline` prices), and its first stage
KERNELS = {"c2": ("pt_samples_q<false, false, 0", "pt_prefix<false, false>"),
           "c3": ("pt_samples_q<false, false, 1", "pt_prefix<false, false>"),
           "c4": ("pt_samples_q<false, true, 0", "pt_prefix<false, true>"),
           "c5": ("pt_samples_w<false>", "pt_prefix<false, true>")}
# crop of the frame compared with the CPU oracle (fractions of the frame: x, y; then width, height in pixels)
CROP = {"c2": (0.469, 0.306, 32, 16), "c3": (0.484, 0.278, 32, 16), "c4": (0.5, 0.25, 8, 4), "c5": (0.495, 0.417, 4, 2)}
# corner of the frame the reference's OpenCL build renders for the parity leg (it searches by brute force)
REAL_OPENCL_CORNER = {"c2": None, "c3": None, "c4": (64, 16), "c5": (32, 8)}
# flops of the reference's routines, per counter of rt_counters: hitSphere miss path 17; hitPlane
# 11; hitLens 40; hitTriangle 27 to its last reject + 25 more when it hits (uv, point, normal); per hit bounce 40
# (point, normal, material, mixCol); rayScatter 15 (+ normalize); the dielectric branch 30; primary ray 25
FLOPS = {"t_sphere": 17, "t_plane": 11, "t_lens": 40, "t_tri": 27, "h_tri": 25, "h_bounce": 40, "n_scatter": 15,
         "n_dielectric": 30, "samples": 25}


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


def profile_key(workload, width, height, spp):
    return "%s@%dx%dx%d" % (workload, width, height, spp)


def self_launch(n_gpus):
    """`python bench.py --gpus N` without a torchrun environment: start the N ranks as a child process (never exec:
    this process may already hold the GPU) and relay its one JSON line and its return code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    print("[bench] --gpus %d without a torchrun environment: starting the ranks myself\n[bench]   %s" %
          (n_gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    sys.stdout.write(p.stdout)
    sys.stdout.flush()
    return p.returncode


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
            "arith": "ieee (the oracle's contract)",
            "sample": "%d band(s) of %d rows x %d columns of the %dx%d frame, %d of %d spp = %.3f M pixel-samples, "
                      "%.1f s wall" % (nb, rows, cw_s, W, H, spp_s, spp, samples / 1e6, wall)}


def roofline(workload, key, arith, main_ms, first_ms, call_ms, counters, share=1.0, valid=True):
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
            mix = json.load(open(path)).get(key)
        except Exception:
            mix = None
    main_key, first_key = KERNELS[workload]
    ns = "pt_a%d::" % ARITH[arith]
    kern = None
    if mix and mix.get("arith") == arith:
        for name, k in mix.get("kernels", {}).items():
            if name.startswith(ns + main_key):
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
        out["note"] = ("profiles/valu_mix.json has no entry '%s' for arithmetic '%s': run tools/profile.sh + "
                       "tools/summarize_profile.py" % (key, arith))
    # work / peak: the reference's logical flops (its brute-force searches included) over the call time.  For a workload
    # that runs through a BVH (C4, C5) most of these flops are never executed: a figure above 1 says "algorithmically
    # avoided", not "faster than the chip".
    flops = float(sum(FLOPS[k] * getattr(counters, k) for k in FLOPS))
    out["alg_flop_frac"] = {"value": round(flops / (call_ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 4),
                            "algorithmic_flops_per_launch": int(flops), "peak_tflops": FP32_PEAK_TFLOPS,
                            "note": "reference's logical fp32 flops (SURVEY 8d item 3: 17 per sphere test, 11 plane, 27 + 25 "
                                    "triangle, 40 per hit bounce, 15 scatter, 30 dielectric, 25 per primary ray) / call time / "
                                    "vector-fp32 peak" + ("; brute-force work a BVH skips is counted: not a physical rate"
                                                          if workload in ("c4", "c5") else "")}
    # round 1's figure, kept for continuity: the REFERENCE kernel's logical traffic (every primitive struct it
    # would have dereferenced) over the call time.  The scene lives in SGPRs / L2, so these bytes never move:
    # not a physical rate, and not bounded by 1.
    alg_bytes = counters.algorithmic_bytes()
    out["alg_hbm_frac"] = {"value": round(alg_bytes / (call_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                           "algorithmic_bytes_per_launch": int(alg_bytes),
                           "note": "non-physical: reference's brute-force logical bytes / time / 8 TB/s (SURVEY 8d)"}
    return out


def parity_leg(rt, np, tracer, wl, workload, spp, arith, table, note):
    """Untimed.  (1) the timed policy against the reference's gfx950 OpenCL code object (the pin to a real OpenCL
    build): samples 0 and 1 of the compared region bit for bit, the timed frame's linear radiance within 1e-4;
    (2) policy 0 against the CPU oracle: probes bit for bit, a crop of a frame within 1e-4."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc_mod
    W, H = wl.width, wl.height
    out = {"arith_timed": arith, "real_opencl": None, "cpu_oracle_ieee": None}
    timed_linear = tracer.readLinear()[..., :3].astype(np.float64)     # the last timed step's accumulator / count
    # ---- (1)
    hs = {"rocm-opencl": orc_mod.REF950_HSACO, "rocm-opencl-nocontract": orc_mod.REF950_HSACO_NOCONTRACT,
          "ieee": orc_mod.REF950_HSACO}[arith]
    if wl.scene.texture_args()[3]:
        out["real_opencl"] = {"skipped": "t_textured materials need an OpenCL image object, which a HIP process cannot create "
                                         "(raytracer.cl:105-107 stays unpinned; tests/test_gpu_ref950.py compares the untextured scene)"}
    elif not orc_mod.ReferenceGfx950.available(hs):
        out["real_opencl"] = {"skipped": "oracle/_ref_gfx950 (the reference built by ROCm's OpenCL tool chain) is not on this machine"}
    else:
        note("parity: against the reference's gfx950 OpenCL build (%s)" % os.path.basename(hs))
        ref = orc_mod.ReferenceGfx950(hs)
        grid = REAL_OPENCL_CORNER[workload]
        gw, gh = grid if grid else (W, H)
        same = []
        for k in (0, 1):
            _, last = ref.render(wl.scene, wl.camera, table, W, H, k, 1, want_last=True, grid=grid)
            tracer.clear()
            tracer.renderSamples(wl.camera, k, 1)
            tracer.sync()
            mine = tracer.readLinear()[:gh, :gw, :3]
            same.append(float((mine.view(np.uint32) == last[:gh, :gw, :3].view(np.uint32)).all(axis=2).mean()))
        a = ref.render(wl.scene, wl.camera, table, W, H, 0, spp, grid=grid)[:gh, :gw, :3].astype(np.float64) / spp
        b = timed_linear[:gh, :gw]
        rel = np.abs(a - b) / np.maximum(np.maximum(np.abs(a), np.abs(b)), 1e-6)
        out["real_opencl"] = {
            "checker": "oracle/_ref_gfx950/" + os.path.basename(hs) + " (kernels/raytracer.cl built by ROCm's OpenCL tool chain, ROCm's builtin library)",
            "region": "whole frame" if not grid else "corner %dx%d of the frame" % (gw, gh),
            "pixel_samples_bit_identical_samples_0_1": same,
            "timed_frame_max_rel_dev": float(rel.max()), "timed_frame_pixels_within_1e-4": float((rel <= 1e-4).mean()),
            "lit_fraction": float((a.sum(axis=2) > 0).mean()),
            "ok": (arith == "ieee") or (min(same) == 1.0 and float(rel.max()) <= 1e-4),
            **({"note": "policy ieee is the CPU oracle's contract and does not claim bit identity with an OpenCL build: distance only"}
               if arith == "ieee" else {})}
    # ---- (2)
    note("parity: policy ieee against the CPU oracle")
    orc = orc_mod.Oracle()
    tracer.setArith("ieee")
    rng = np.random.RandomState(1)
    n = 64 if workload in ("c4", "c5") else 2000
    xs, ys, ss = rng.randint(0, W, n), rng.randint(0, H, n), rng.randint(0, spp, n)
    got = tracer.traceSamples(wl.camera, xs, ys, ss)
    exp, _ = orc.samples(wl.scene, wl.camera, table, W, H, xs, ys, ss)
    bit = int((got.view(np.uint32) == exp.view(np.uint32)).all(axis=1).sum())
    fx, fy, cw, ch = CROP[workload]
    x0, y0 = min(int(fx * W), W - cw), min(int(fy * H), H - ch)
    frame = tracer.renderFrame(wl.camera, spp)
    ref_img, _ = orc.render(wl.scene, wl.camera, table, W, H, 2, count=spp, region=(x0, y0, cw, ch), threads=host_cores())
    a, b = frame[y0:y0 + ch, x0:x0 + cw], ref_img[y0:y0 + ch, x0:x0 + cw]
    dev = float((np.abs(a - b) / np.maximum(np.maximum(np.abs(a), np.abs(b)), 1e-6)).max())
    out["cpu_oracle_ieee"] = {"checker": "oracle/pt_oracle.c", "probes": n, "bit_exact": bit, "frame_crop": [x0, y0, cw, ch],
                              "crop_max_rel_dev": dev, "crop_ok": dev <= 1e-4,
                              "crop_lit_fraction": float((b[..., :3].sum(-1) > 0).mean())}
    tracer.setArith(arith)
    out["walk_overflow"] = tracer.walkOverflow()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOAD_DESC),
                    help="default c2, the configuration the metric is quoted on — at every N")
    ap.add_argument("--scaling", default="auto", choices=["auto", "strong", "weak"],
                    help="N > 1: auto = weak for c2 / c3 (64 x N spp in one call per rank), strong for c4 / c5 (the same frame sharded)")
    ap.add_argument("--size", default="", metavar="WxH", help="override the frame size, e.g. 1920x1080")
    ap.add_argument("--spp", type=int, default=0, help="override samples per pixel")
    ap.add_argument("--arith", default="rocm-opencl", choices=sorted(ARITH),
                    help="arithmetic policy of the timed kernels (csrc/pt_arith.hpp)")
    ap.add_argument("--workload-arg", action="append", default=[], metavar="KEY=INT",
                    help="rehearsals only: override a generator argument of the workload, e.g. n_spheres=20000")
    ap.add_argument("--exchange", default="gather", choices=["gather", "reduce"],
                    help="N>1: gather of packed tiles (default) or full-frame RCCL reduce")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity-check", action="store_true")
    ap.add_argument("--no-single-gpu-reference", action="store_true")
    ap.add_argument("--no-arith-variants", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=20.0, help="seconds of CPU work for the baseline leg (N > 1: a quarter)")
    args = ap.parse_args()

    # a plain `python bench.py --gpus N`: start the ranks as a child BEFORE anything touches torch or HIP
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))

    import numpy as np
    import torch
    import opencl_raytracing_amd as rt
    from importlib import import_module
    dist_mod = import_module("opencl-raytracing_amd.distributed")

    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: no HIP device visible (the product has no CPU path)")
    rank, world, local = dist_mod.init_process_group()
    if world != args.gpus:
        sys.exit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run --nproc-per-node %d, or plain)" %
                 (args.gpus, world, args.gpus))
    device = local % torch.cuda.device_count()   # == local on a full node; rehearsals may oversubscribe a GPU
    torch.cuda.set_device(device)
    import torch.distributed as dist

    workload = args.workload
    scaling = args.scaling if args.scaling != "auto" else ("weak" if workload in ("c2", "c3") else "strong")
    wl_args = {k: int(v) for k, v in (a.split("=", 1) for a in args.workload_arg)}
    if args.size:
        w_, h_ = args.size.lower().split("x")
        wl_args.update(width=int(w_), height=int(h_))
    wl = rt.workloads.get(workload, **wl_args)
    generator_overridden = any(k not in ("width", "height") for k in wl_args)
    base_spp = args.spp or wl.spp
    wl.spp = base_spp
    spp = base_spp * world if scaling == "weak" else base_spp
    # samples per fused call: ONE call per step up to 512 samples per pixel (the capacity of a wave's sample queue); beyond
    # that calls of 512 when that divides the count (the library itself would split a larger call the same way)
    chunk = spp if spp <= 512 else (512 if spp % 512 == 0 else base_spp)
    key = profile_key(workload, wl.width, wl.height, base_spp)
    tracer = rt.RayTracer(wl.width, wl.height, scene=wl.scene, device=device, seed=rt.workloads.SEED)
    tracer.setArith(args.arith)
    tracer.resetCounters()
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

    # algorithmic work of one launch: exact work counters from the counting build (untimed)
    note("counting pass (%s, %s, %d rank(s), %s scaling, arithmetic %s)" % (workload, key, world, scaling, args.arith))
    tracer.enableCounters(True)
    step()
    tracer.sync()
    cn = tracer.counters()
    tracer.enableCounters(False)
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

    # the same workload on ONE GPU, by rank 0, outside the timed region: the frame of the N = 1 line
    single = None
    if world > 1 and not args.no_single_gpu_reference:
        if rank == 0:
            note("single-GPU reference of the same workload (untimed leg)")
            solo = rt.RayTracer(wl.width, wl.height, scene=wl.scene, device=device, seed=rt.workloads.SEED)
            solo.setArith(args.arith)
            n_solo = 3 if workload in ("c4", "c5") else 10
            solo.renderFrameOnDevice(wl.camera, base_spp)
            solo.sync()
            t1 = time.perf_counter()
            for _ in range(n_solo):
                solo.renderFrameOnDevice(wl.camera, base_spp)
            solo.sync()
            single = (time.perf_counter() - t1) / n_solo * 1e3
            solo.close()
        dist.barrier()

    if rank == 0:
        ms_per_step = wall_max / args.steps * 1e3
        value = total_samples * args.steps / wall_max / 1e6
        roof = roofline(workload, key, args.arith, float(np.mean(main_ms)), float(np.mean(first_ms)), call_ms, cn,
                        share=my_samples / float(wl.width * wl.height * base_spp) / (spp // chunk),
                        valid=not generator_overridden)
        out = {
            "metric": "Msamples/sec (rays/sec) at %d×%d, %d spp; %% HBM roofline" % (wl.width, wl.height, spp),
            "value": round(value, 2), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "hbm_roofline_pct": round(100.0 * roof["hbm_frac"], 3) if roof.get("hbm_frac") is not None else None,
            "arith": args.arith,
            "config": {"workload": "%s, %dx%d, %d spp" % (WORKLOAD_DESC[workload], wl.width, wl.height, spp),
                       "width": wl.width, "height": wl.height, "spp_per_call": chunk, "spp_total": spp,
                       "sharding": "8x8 tiles interleaved over %d rank(s)" % world +
                                   ("; %s of the packed radiance tiles to rank 0 over RCCL" % args.exchange if world > 1 else ""),
                       "pixel_samples_per_step": int(total_samples), "seed": hex(rt.workloads.SEED),
                       "profile_key": key,
                       "arithmetic": "%s (RT_OPT_ARITH %d, csrc/pt_arith.hpp)" % (args.arith, ARITH[args.arith]),
                       **({"workload_overrides": wl_args} if wl_args else {})},
            "roofline": roof,
            "kernel_ms_min_max": [round(min(ev_ms), 4), round(max(ev_ms), 4)],
            "bounces_per_sample": round(cn.bounces / max(cn.samples, 1), 3),
        }
        if single is not None:
            # strong: the same frame → time ratio; weak: N frames' worth of pixel-samples against one → throughput ratio
            speedup = single / ms_per_step * (world if scaling == "weak" else 1)
            out["single_gpu_same_workload"] = {
                "ms_per_step": round(single, 4),
                "value": round(wl.width * wl.height * base_spp / (single * 1e-3) / 1e6, 2),
                "note": "rank 0 alone, the %d-spp frame of the N = 1 line, same run, outside the timed region" % base_spp}
            out["speedup_vs_1gpu_same_workload"] = round(speedup, 3)
        if world == 1 and not args.no_arith_variants:
            # the same frame under the other arithmetic policies (5 steps each, untimed leg): what the policy costs
            variants = {}
            for name in sorted(ARITH, key=ARITH.get):
                tracer.setArith(name)
                for _ in range(6):
                    step()
                tracer.sync()
                f_ms, m_ms = tracer.stageMsHistory(5)
                variants[name] = {"first_stage_ms": round(float(np.mean(f_ms)), 4), "kernel_ms": round(float(np.mean(m_ms)), 4)}
            tracer.setArith(args.arith)
            step()                      # the accumulator holds a frame of the timed policy again
            tracer.sync()
            out["arith_variants"] = variants
        if world == 1 and not args.no_parity_check:
            out["parity"] = parity_leg(rt, np, tracer, wl, workload, spp, args.arith, table, note)
        else:
            out["walk_overflow"] = tracer.walkOverflow()
        if not args.no_cpu_baseline:
            note("cpu baseline")
            out["cpu_baseline"] = cpu_baseline(wl, table, args.cpu_budget if world == 1 else args.cpu_budget / 4)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    tracer.close()


if __name__ == "__main__":
    main()
