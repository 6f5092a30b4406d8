#!/bin/bash
# Dev helper (GPU box): round 4's randomised parity campaigns on the final build.  usage: r04_campaign.sh <first seed> <seconds each>
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/campaign; mkdir -p $O
python scripts/parity_campaign.py 6000 ${1:-40000000} ${2:-500} 2>/dev/null | tee $O/mixed_$1.txt | grep -v "^\.\.\. .*[05]0 cases" 
MRT_CAMPAIGN_LARGE=1 python scripts/parity_campaign.py 6000 $(( ${1:-40000000} + 1000000 )) ${2:-500} 2>/dev/null | tee $O/large_$1.txt | grep -v "^\.\.\. .*[05]0 cases"
true
