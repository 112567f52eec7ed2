#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$PWD}
for a in "0 64" "2 64" "1 64" "0 128" "2 128" "1 128"; do bash tools/mfma_trace.sh $a 2>&1 | grep scan_mfma; done
