#!/bin/bash
# one_pass of the trees under _ab/ and of the working tree, alternating (three passes over the list) on one box
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for t in "$@"; do python tools/one_pass_probe.py $t 30 2>&1 | grep "^one_pass"; done
done
