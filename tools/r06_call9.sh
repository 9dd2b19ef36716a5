#!/bin/bash
# round 6, call 9: deferred activations above K = 256 re-measured with the packed transform (VERDICT r5 Next 1d)
export BENCH_ARGS="--steps 12 --warmup 3 --no-second-line --no-knn-check"
bash tools/ab_env.sh "k256:CCN_LAZY_ACT_MAX_K=256" "k512:CCN_LAZY_ACT_MAX_K=512" "k1024:CCN_LAZY_ACT_MAX_K=1024" "k256b:CCN_LAZY_ACT_MAX_K=256"
