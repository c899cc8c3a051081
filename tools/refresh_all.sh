#!/bin/bash
# tools/refresh_all.sh <round tag> — GPU box, ONE gpurun call: rocprofv3 passes of every workload (tools/profile_all.sh),
# their summaries (tools/summarize_profile.py → profiles/*_summary.md, valu_mix.json, traffic.json), then the bench lines
# against that fresh instruction mix (tools/bench_all.sh), and everything copied to gpurun_out/profiles_out/ — the only
# directory that travels back.  Afterwards, in the build container:  cp gpurun_out/profiles_out/* profiles/
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
bash tools/profile_all.sh $TAG || exit 1
summ() { python3 tools/summarize_profile.py "$@" > /dev/null || { echo "summary $1 failed"; exit 1; }; }
summ ${TAG}_c2 c2 "pt_prefix<false, false>;pt_samples_q<false, false, 0, 6>;pt_tree_pass<false>"
summ ${TAG}_c3_64 c3 "pt_prefix<false, false>;pt_samples_q<false, false, 1, 6>;pt_tree_pass<false>"
summ ${TAG}_c3 c3 "pt_prefix<false, false>;pt_samples_q<false, false, 1, 6>;pt_tree_pass<false>"
summ ${TAG}_c4 c4 "pt_prefix<false, true>;pt_samples_q<false, true, 0, 6>;pt_tree_pass<true>"
summ ${TAG}_c5_1080p_64 c5 "pt_prefix<false, true>;pt_samples_w<false>;pt_tree_pass<true>"
summ ${TAG}_c5 c5 "pt_prefix<false, true>;pt_samples_w<false>;pt_tree_pass<true>"
cp profiles/${TAG}_c2_summary.md profiles/${TAG}_summary.md
cp profiles/${TAG}_c2_kernel_stats.csv profiles/${TAG}_kernel_stats.csv
bash tools/bench_all.sh $TAG
mkdir -p gpurun_out/profiles_out
for n in c2 c3_64 c3 c4 c5_1080p_64 c5; do cp gpurun_out/${TAG}_bench_$n.json profiles/${TAG}_bench_$n.json; done
cp profiles/${TAG}_*summary.md profiles/${TAG}_*kernel_stats.csv profiles/${TAG}_bench_*.json profiles/valu_mix.json profiles/traffic.json gpurun_out/profiles_out/
