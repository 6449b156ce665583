#!/bin/bash
# Round-3 profile set (GPU box): kernel trace + stats, then separate PMC passes, for the shapes bench.py times.
#   r03_lone   one lone 1024-pair launch per step on one stream (bench.py kernel_only)   -> the dominant kernel's own duration
#   r03_b42    4 streams x 2 launches of 512 pairs per step, three waves per SIMD
#   r03_b21    2 streams x 1 launch of 1024 pairs per step, three waves per SIMD
#   r03_c3     config 3 at the benched shape: 1024 profile pairs 2000x2000 (tools/bench_c3.py)
#   r03_c4     config 4 (tools/bench_c4.py 1024 2000 256 0.01 cw 10)
#   r03_c5     config 5 slice (tools/bench_c5.py 4096 512)
# usage: tools/profile_r03.sh [tags...]   (default: all)
set -e
cd "$GRAFT_REPO_ROOT"
TAGS="${@:-lone b42 b21 c3 c4 c5 sec}"
B="--steps 10 --warmup 3 --no-secondary --no-cpu-baseline --lone-steps 0"
for t in $TAGS; do
  case $t in
    lone) tools/profile.sh r03_lone -- python3 bench.py $B --streams 1 --split 1 --occupancy 2 --alt-prio 1 ;;
    b42)  tools/profile.sh r03_b42 -- python3 bench.py $B --streams 4 --split 2 --occupancy 3 --alt-prio 1 ;;
    b21)  tools/profile.sh r03_b21 -- python3 bench.py $B --streams 2 --split 1 --occupancy 3 --alt-prio 1 ;;
    c3)   tools/profile.sh r03_c3 -- python3 tools/bench_c3.py 1024 2000 1 ;;
    c4)   tools/profile.sh r03_c4 -- python3 tools/bench_c4.py 1024 2000 256 0.01 cw 10 ;;
    c5)   tools/profile.sh r03_c5 -- python3 tools/bench_c5.py 4096 512 ;;
    sec)  tools/profile.sh r03_sec -- python3 tools/bench_secondary.py 1024 2000 ;;
  esac
  echo "profiled $t"
done
