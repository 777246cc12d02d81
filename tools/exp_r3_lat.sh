#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
for v in "A=1" "HC_SIDE_PRIO=0" "A=2"; do
  echo "== $v"; env $v timeout -k 10 200 python tools/exp_latency.py 2>/dev/null | grep -E "detectMultiScale|detectFaces|batch of 1"
done
