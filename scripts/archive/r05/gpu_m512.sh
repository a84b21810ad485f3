#!/bin/bash
# 20-step protocol against the 2000-step one: queued-eager launches behind 2000 untimed launches vs one graph replay
cd $GRAFT_REPO_ROOT
for a in "--steps 20 --warmup 5" "--steps 20 --warmup 5 --timed-launch graph" "--steps 200 --warmup 5" "--steps 2000 --warmup 50"; do
  echo "== $a"
  timeout 600 python bench.py --cpu-seconds 0 --decode-seconds 0 --layers "" $a 2>&1 >/dev/null | grep "M= "
done
