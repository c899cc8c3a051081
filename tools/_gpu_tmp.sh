cd $GRAFT_REPO_ROOT
for i in 1 2; do
timeout -k 10 200 python3 tools/quick_bench.py c2:64 2>/dev/null
timeout -k 10 200 python3 tools/quick_bench.py --lib=opencl-raytracing_amd/variants/pre1.so c2:64 2>/dev/null
timeout -k 10 200 python3 tools/quick_bench.py --lib=opencl-raytracing_amd/variants/pre2.so c2:64 2>/dev/null
done
