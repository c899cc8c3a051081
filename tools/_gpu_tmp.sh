set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 200 python3 tools/quick_bench.py c5:16:960:540 c5:64:1920:1080 2>/dev/null || exit 1
timeout -k 10 200 python3 tools/quick_bench.py --lib=opencl-raytracing_amd/variants/wbw4.so c5:16:960:540 c5:64:1920:1080 2>/dev/null || exit 1
