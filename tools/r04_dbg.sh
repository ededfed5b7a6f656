#!/usr/bin/env bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; bash tools/pmc_latency.sh r04z/pmc_lat
