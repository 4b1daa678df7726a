#!/bin/bash
# Builds (if needed) and runs the attention lab on the GPU box: bash tools/attn_lab/run.sh [variant ...]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R/tools/attn_lab
[ -x attn_lab ] || bash build.sh || exit 1
timeout 600 ./attn_lab "$@"
