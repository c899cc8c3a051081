set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 200 python3 tools/quick_bench.py c2:64 c3:256 c4:8 c4:64 c5:16:960:540 c5:64:1920:1080 2>/dev/null || exit 1
timeout -k 10 200 python3 tools/quick_bench.py --lib=opencl-raytracing_amd/variants/q7.so c2:64 c3:256 2>/dev/null || exit 1
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu --deselect tests/test_gpu_bench.py 2>&1 | tail -4
