set -o pipefail
cd $GRAFT_REPO_ROOT
python3 bench.py > gpurun_out/r02_bench_c2.json 2> gpurun_out/r02_bench_c2.err || { tail -5 gpurun_out/r02_bench_c2.err; exit 1; }
python3 bench.py --workload c3 --steps 10 > gpurun_out/r02_bench_c3.json 2> gpurun_out/r02_bench_c3.err || { tail -5 gpurun_out/r02_bench_c3.err; exit 1; }
python3 bench.py --workload c4 --steps 5 --warmup 1 > gpurun_out/r02_bench_c4.json 2> gpurun_out/r02_bench_c4.err || { tail -5 gpurun_out/r02_bench_c4.err; exit 1; }
python3 bench.py --workload c5 --steps 3 --warmup 1 > gpurun_out/r02_bench_c5.json 2> gpurun_out/r02_bench_c5.err || { tail -5 gpurun_out/r02_bench_c5.err; exit 1; }
timeout -k 10 900 python3 -m pytest tests/test_gpu_bench.py tests/test_gpu_ref950.py -x -q -m gpu 2>&1 | tail -3
