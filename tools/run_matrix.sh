#!/bin/bash
# the parity, fuzz, cull and multi-rank tests under every product-path switch (one summary line per switch):
# scene options (prt_scene_options) and trace flags (PRT_TRACE_*), handed to the product by tests/conftest.py
for env in "PRT_TEST_FLAGS=1" "PRT_TEST_FLAGS=2" "PRT_TEST_FLAGS=4" "PRT_TEST_FLAGS=8" "PRT_TEST_FLAGS=16" "PRT_TEST_FLAGS=1024" "PRT_TEST_OPTIONS=no_chain=1" "PRT_TEST_OPTIONS=no_cull=1" \
           "PRT_TEST_OPTIONS=no_groups=1" "PRT_TEST_OPTIONS=list_order_groups=1" "PRT_TEST_OPTIONS=no_implied=1" \
           "PRT_TEST_OPTIONS=hit_lanes=8" "PRT_TEST_OPTIONS=no_intervals=1" "PRT_TEST_OPTIONS=no_clearance=1"; do
  echo "== $env"
  env $env PRT_FUZZ_SEEDS=${PRT_FUZZ_SEEDS:-200} python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_cull.py tests/test_gpu_distributed.py tests/test_gpu_record_plan.py -m gpu -q -x -k "not dense_mode and not state_rows and not hint and not plan_arguments" 2>&1 | grep -E "passed|failed|FAILED" | tail -3
done
