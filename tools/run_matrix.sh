#!/bin/bash
# the parity, fuzz, cull and multi-rank tests under every product-path switch (one summary line per switch)
for env in "PRT_NO_HINTS=1" "PRT_FULL_ROWS=1" "PRT_PUBLISH_KERNEL=1" "PRT_NO_CHAIN=1" "PRT_NO_CULL=1" "PRT_NO_GROUPS=1" "PRT_HIT_VARIANT=lanes8"; do
  echo "== $env"
  env $env PRT_FUZZ_SEEDS=200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_cull.py tests/test_gpu_distributed.py -m gpu -q -x -k "not dense_mode and not state_rows and not hint" 2>&1 | grep -E "passed|failed|FAILED" | tail -3
done
