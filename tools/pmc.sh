#!/bin/bash
# usage (on the GPU box): tools/pmc.sh <tag> "<counters>" <quick_bench spec...>
TAG=$1; shift; CTRS=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CTRS --output-format csv -d $ROOT/gpurun_out/pmc_$TAG -- python3 $ROOT/tools/quick_bench.py "$@" > /dev/null 2>&1
python3 - <<PY
import csv,glob,collections
d=collections.defaultdict(list)
for f in glob.glob("$ROOT/gpurun_out/pmc_$TAG/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "pt_" in r["Kernel_Name"]:
            d[(r["Kernel_Name"].split("(")[0][:28], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k in sorted(d): print("%-30s %-26s %.4g" % (k[0], k[1], sum(d[k])/len(d[k])))
PY
