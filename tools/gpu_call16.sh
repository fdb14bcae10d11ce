#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 300 python tools/step_timeline.py 256 2>&1 | tail -22
