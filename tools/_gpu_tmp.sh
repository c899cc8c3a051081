cd $GRAFT_REPO_ROOT
for v in slot3 slot1b slot1 slot1c; do timeout -k 10 200 python3 tools/quick_bench.py --lib=opencl-raytracing_amd/variants/$v.so c2:64 c3:256 2>/dev/null || echo "$v failed"; done
