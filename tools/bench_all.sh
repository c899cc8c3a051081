#!/bin/bash
# tools/bench_all.sh <tag> [extra bench args] — GPU box: one bench line per workload of the round's table → gpurun_out/<tag>_<name>.json
TAG=${1:-r03}; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT && mkdir -p gpurun_out
run() { name=$1; shift; timeout -k 10 400 python3 bench.py "$@" > gpurun_out/${TAG}_bench_$name.json 2> gpurun_out/${TAG}_bench_$name.err || echo "bench $name failed"; python3 - <<PY
import json
try:
    j = json.loads([l for l in open("gpurun_out/${TAG}_bench_$name.json") if l.startswith("{")][-1])
    r = j["roofline"]
    print("%-12s %10.1f Msamples/s  %9.4f ms/step  kernel %9.4f ms  prefix %.4f  frac %s  arith %s  variants %s" % ("$name", j["value"], j["ms_per_step"], r["kernel_ms"], r["first_stage_ms"], r.get("frac"), j["arith"], {k: v["kernel_ms"] for k, v in j.get("arith_variants", {}).items()}))
    p = j.get("parity") or {}
    print("             parity real_opencl:", (p.get("real_opencl") or {}).get("ok"), (p.get("real_opencl") or {}).get("pixel_samples_bit_identical_samples_0_1"), "cpu_oracle_ieee:", (p.get("cpu_oracle_ieee") or {}).get("bit_exact"), (p.get("cpu_oracle_ieee") or {}).get("crop_ok"), "overflow:", p.get("walk_overflow"))
except Exception as e:
    print("$name: no line", e)
PY
}
run c2 --steps 20 --warmup 3 "$@"
run c3_64 --workload c3 --spp 64 --steps 10 --warmup 2 "$@"
run c3 --workload c3 --steps 5 --warmup 1 "$@"
run c5_1080p_64 --workload c5 --size 1920x1080 --spp 64 --steps 5 --warmup 1 "$@"
run c4 --workload c4 --steps 3 --warmup 1 "$@"
run c5 --workload c5 --steps 2 --warmup 1 "$@"
