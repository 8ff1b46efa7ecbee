#!/bin/bash
# usage: tools/pmc.sh <outdir> <counters...> -- <program and args>     (run through gpurun, on the GPU box)
# Collects PMC counters in a pass of their own (kernel-trace only; the program itself directly after `--`)
# and prints per-kernel means of the acfm kernels.
set -o pipefail
[ -n "$GRAFT_REPO_ROOT" ] || { echo "GRAFT_REPO_ROOT is not set (run through gpurun)"; exit 2; }
out=$1; shift
ctrs=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do ctrs+=("$1"); shift; done
shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 2
rm -rf "$out"
rocprofv3 --pmc "${ctrs[@]}" --kernel-trace --output-format csv -d "$out" -- "$@" > "$out.log" 2>&1 || { tail -5 "$out.log"; exit 1; }
python3 tools/pmc_summary.py "$out"
