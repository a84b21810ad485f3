#!/bin/bash
# after a planner change: tests + the small / 17..64 / 24..448 / generalisation audits in one session
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -3
bash scripts/gpu_planner_sweep.sh
bash scripts/gpu_planner_sweep_17_64.sh
bash scripts/gpu_planner_sweep_mid.sh
bash scripts/gpu_planner_sweep_gen.sh
bash scripts/gpu_planner_sweep_large.sh
