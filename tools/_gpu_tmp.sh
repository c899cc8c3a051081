set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 200 python3 tools/quick_bench.py c3:256 c5:16:960:540 c5:64:1920:1080 2>/dev/null || exit 1
for v in le1 le2 le8; do timeout -k 10 200 python3 tools/quick_bench.py --lib=opencl-raytracing_amd/variants/$v.so c5:64:1920:1080 2>/dev/null || echo "$v failed"; done
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu --deselect tests/test_gpu_bench.py 2>&1 | tail -4
