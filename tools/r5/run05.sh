#!/bin/bash
cd "$(dirname "$0")/../.."
for i in 1 2 3; do python -m pytest tests/test_models_gpu.py -q -m gpu -x -k "launch_plans_replay_matches_eager and linknet34" 2>&1 | grep -E "assert err|passed|failed|^E  +AssertionError" | head -5; done
