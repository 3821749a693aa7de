#!/bin/bash
# All bench workloads through tools/gpu_pmc.sh, each with the kernel the library's measured choice ends up with pinned
# (so that a profile holds ONE kernel): tools/gpu_pmc_all.sh TAG [small|soup]
set -u
TAG=${1:-r03}; WHICH=${2:-all}
if [ "$WHICH" != soup ]; then
RT_COOP=0 RT_TRAVERSAL=2 bash tools/gpu_pmc.sh teapot_4k_d1 ${TAG}_teapot4k 12 || exit 1
RT_COOP=0 RT_TRAVERSAL=2 bash tools/gpu_pmc.sh teapot_1080p_d1 ${TAG}_teapot1080p 12 || exit 1
RT_COOP=1 RT_TRAVERSAL=2 bash tools/gpu_pmc.sh testscene1_720p_d10 ${TAG}_testscene1 20 || exit 1
RT_COOP=1 RT_TRAVERSAL=2 bash tools/gpu_pmc.sh teapottable_1080p_d8 ${TAG}_teapottable 12 || exit 1
fi
if [ "$WHICH" != small ]; then
RT_TRAVERSAL=3 bash tools/gpu_pmc.sh soup100k_1080p_d4 ${TAG}_soup100k 12 || exit 1
RT_TRAVERSAL=3 bash tools/gpu_pmc.sh soup1m_4k_d4 ${TAG}_soup1m 8 || exit 1
fi
