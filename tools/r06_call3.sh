#!/bin/bash
# round 6, call 3: weight gradients beside the backward pass -- side stream with ONE weight-gradient workgroup per CU (VERDICT r5 Next 1e)
export BENCH_ARGS="--steps 16 --warmup 3 --no-second-line --no-knn-check"
bash tools/ab_env.sh "base:CCN_X=0" "ws_bg:CCN_WGRAD_STREAM=1 CCN_WGRAD_BG=86016" \
   "ws_bg_hi:CCN_WGRAD_STREAM=1 CCN_WGRAD_BG=86016 CCN_BENCH_MAIN_PRIORITY=-1" "bg_only:CCN_WGRAD_BG=86016" "ws_hi:CCN_WGRAD_STREAM=1 CCN_BENCH_MAIN_PRIORITY=-1" "base2:CCN_X=0"
