#!/bin/bash
# tools/profile_all.sh <round tag> — GPU box: rocprofv3 kernel trace + PMC passes (tools/profile.sh) of every workload of the
# round's table → gpurun_out/prof_<tag>_<name>/; summaries: tools/summarize_profile.py <tag>_<name> <workload> "<kernels>"
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
bash tools/profile.sh ${TAG}_c2 > gpurun_out/profile_${TAG}_c2.log 2>&1; echo "c2 rc=$?"
bash tools/profile.sh ${TAG}_c3_64 --workload c3 --spp 64 > gpurun_out/profile_${TAG}_c3_64.log 2>&1; echo "c3_64 rc=$?"
bash tools/profile.sh ${TAG}_c3 --workload c3 > gpurun_out/profile_${TAG}_c3.log 2>&1; echo "c3 rc=$?"
bash tools/profile.sh ${TAG}_c5_1080p_64 --workload c5 --size 1920x1080 --spp 64 > gpurun_out/profile_${TAG}_c5_1080p_64.log 2>&1; echo "c5_1080p_64 rc=$?"
bash tools/profile.sh ${TAG}_c4 --workload c4 > gpurun_out/profile_${TAG}_c4.log 2>&1; echo "c4 rc=$?"
bash tools/profile.sh ${TAG}_c5 --workload c5 > gpurun_out/profile_${TAG}_c5.log 2>&1; echo "c5 rc=$?"
