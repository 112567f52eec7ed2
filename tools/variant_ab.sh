#!/bin/bash
# timing of 64/128-query sweeps per library variant ($@ = variant names; "default" = the shipped library)
cd ${GRAFT_REPO_ROOT:-$PWD}
for v in "$@"; do for nq in 64 128; do bash tools/mfma_trace.sh $v $nq 2>&1 | grep variant; done; done
