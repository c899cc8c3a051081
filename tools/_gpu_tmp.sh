set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python3 tools/ref_distance.py --hsaco default --out gpurun_out/ref_distance_default.json > /dev/null 2> gpurun_out/ref_distance_default.err || { tail -5 gpurun_out/ref_distance_default.err; exit 1; }
timeout -k 10 400 python3 tools/ref_distance.py --hsaco nocontract --out gpurun_out/ref_distance_nocontract.json > /dev/null 2> gpurun_out/ref_distance_nocontract.err || { tail -5 gpurun_out/ref_distance_nocontract.err; exit 1; }
echo ok
