#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r05f; mkdir -p $O
( python scripts/queue_oversubscription.py; GPU_MAX_HW_QUEUES=16 python scripts/queue_oversubscription.py; GPU_MAX_HW_QUEUES=24 python scripts/queue_oversubscription.py ) 2>&1 | grep -v amdgpu.ids | tee $O/queues.txt
