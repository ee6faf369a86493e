#!/bin/bash
# k_cull rounds per workgroup under an azimuth shard (LS_CULL_SHARD_ROUNDS, experimental builds): ranks 0, 4, 5 of 8 at SYN-10M, graphs
cd "$(dirname "$0")/.."
E=$PWD/build/exp/$1/liblidarshooter_hip.so
export LS_LIB_PATH=$E LD_PRELOAD=$E
for r in 1 2; do for rounds in ${ROUNDS:-4 8 2}; do
  echo "rounds=$rounds: $(LS_CULL_SHARD_ROUNDS=$rounds W=${W:-syn128x10m} RANKS=0,4,5 MODES=graph timeout -k 10 300 python tools/shard_cost.py 2 8 all 2>&1 | grep 'world 8' | sed 's/.*rank \([0-9]\).*three as frame graphs \([0-9.]*\) us.*/rank \1: \2/' | tr '\n' ' ')"
done; done
