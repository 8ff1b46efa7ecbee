#!/bin/bash
# usage: tools/pmc.sh <outdir> <counters...> -- <kbench args>
# Collects PMC counters (own pass, kernel-trace only) for tools/kbench.py and prints per-kernel means.
out=$1; shift
ctrs=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do ctrs+=("$1"); shift; done
shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --pmc "${ctrs[@]}" --kernel-trace --output-format csv -d "$out" -- python tools/kbench.py "$@" > "$out.log" 2>&1
python tools/pmc_summary.py "$out"
