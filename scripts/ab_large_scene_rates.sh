#!/bin/bash
# Dev helper (GPU box): A/B of a large-scene kernel change -- the GPU tests, then the large-scene rates of the product library and of a
# second build of the library under myraytracer_amd/lib/ (the previous commit's: git worktree add /tmp/wt HEAD && make there), same commands.
# Round 5: 32- vs 24-byte boxes (profiles/r05_boxes_24_vs_32_bytes.txt), 24-byte boxes vs 80-byte sibling blocks (r05_boxes_20_vs_24_bytes.txt).
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r05l; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/tests.txt 2>&1; echo "tests rc=$?" | tee -a $O/tests.txt
tail -n 4 $O/tests.txt
run() { MRT_SHARD=$1 MRT_HINT=$2 MRT_WARMUP=$3 timeout -k 10 300 python scripts/wall_rate.py $5 $6 $7 $8 $4 2>&1 | grep -v amdgpu.ids | sed -e 's/HIER=None BOXES=None RNG=None//' -e "s/^/shard $1 hint $2: /" | cut -c1-120; }
rates() {
  MRT_SHARD= run "" 2,2 8 8 stress 1920 1080 4096
  run 0,8 8,2 32 48 stress 1920 1080 4096
  MRT_SHARD= run "" 2,2 12 40 stress 1920 1080 512
  MRT_SHARD= run "" 2,2 12 80 stress70 1920 1080 64
  MRT_SHARD= run "" 2,2 12 80 stress50 1920 1080 64
  MRT_SHARD= run "" 2,2 12 80 stress36 1920 1080 64
}
( echo "--- 80-byte sibling blocks (bf16 extents)"; rates; echo "--- 24-byte boxes (the previous commit's library)"; MRT_LIB_OVERRIDE=$PWD/myraytracer_amd/lib/libmyraytracer_amd_boxes24.so rates
  echo "--- 80-byte sibling blocks again"; MRT_SHARD= run "" 2,2 8 8 stress 1920 1080 4096; MRT_SHARD= run "" 2,2 12 40 stress 1920 1080 512 ) > $O/rates.txt 2>&1
cat $O/rates.txt
