set -o pipefail
cd $GRAFT_REPO_ROOT
for v in rp1 rp2 rf4 rf16 b8; do timeout -k 10 200 python3 tools/quick_bench.py --lib=opencl-raytracing_amd/variants/$v.so c2:64 c4:8 2>/dev/null || echo "$v failed"; done
