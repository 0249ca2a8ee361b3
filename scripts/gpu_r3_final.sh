#!/bin/bash
# the round's evidence in one call: GPU test suite, bench lines + kernel traces (default, stress), global BA timing + trace
tag=${1:-r3final}
bash scripts/gpu_r3_tests.sh $tag && bash scripts/gpu_r3_bench.sh $tag default stress && bash scripts/gpu_r3_global.sh $tag
