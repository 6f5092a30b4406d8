#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r05h; mkdir -p $O
MRT_TRACE_WIDTH=1 MRT_ONLY=c1_n1,c2_n1,c3_n1,c4_n8,interactive_n1,c3_n8 timeout -k 10 600 python scripts/settle_schedules.py $O/schedules_a.json 3 > $O/settle_a.txt 2>&1
MRT_TRACE_WIDTH=1 MRT_ONLY=c5_n1,c5_n2,c4_n1 timeout -k 10 600 python scripts/settle_schedules.py $O/schedules_b.json 1 > $O/settle_b.txt 2>&1
grep -v amdgpu $O/settle_a.txt $O/settle_b.txt | cut -c1-260
