#!/bin/bash
# Dev helper (GPU box): randomised parity campaigns, HIP vs oracle bit for bit.  usage: campaign.sh <first seed> <seconds each> [tag]
# The campaign's own flushed progress lines go straight to the files under gpurun_out/ (nothing sits in a pipe's buffer: a
# stalled case shows as a file that stops growing, and -- since round 5 -- fails with MRT_ERR_STALLED instead of hanging).
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/campaign; mkdir -p $O
S=${1:-50000000}; T=${2:-400}
MRT_WAIT_TIMEOUT_S=60 python scripts/parity_campaign.py 100000 $S $T > $O/mixed_$S.txt 2>&1; echo "mixed rc=$?"; tail -n 2 $O/mixed_$S.txt
MRT_WAIT_TIMEOUT_S=60 MRT_CAMPAIGN_LARGE=1 python scripts/parity_campaign.py 100000 $(( S + 1000000 )) $T > $O/large_$S.txt 2>&1; echo "large rc=$?"; tail -n 2 $O/large_$S.txt
true
