#!/bin/bash
# A/B of two library builds on ONE GPU box (every gpurun call gets a different MI355X; their bench values differ by up to
# ~3 %, more than most single changes are worth). Run here, in the build container:
#     tools/gpu/ab_build.sh <git-ref>          # builds <git-ref>'s libmolnextr_hip.so into tools/ab/ (git-ignored, ships with gpurun)
#     gpurun --timeout 900 -- 'bash tools/gpu/ab_run.sh [bench args]'
# ab_run.sh alternates current / previous library twice and prints the bench value of each run.
set -e
REF=${1:?usage: ab_build.sh <git-ref>}
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
WT=$(mktemp -d /tmp/mnx_ab.XXXXXX)
git -C "$ROOT" worktree add -q --detach "$WT" "$REF"
make -C "$WT/molnextr_amd/csrc" -j8 > /dev/null
mkdir -p "$ROOT/tools/ab"
cp "$WT/molnextr_amd/lib/libmolnextr_hip.so" "$ROOT/tools/ab/libmolnextr_hip_prev.so"
git -C "$ROOT" worktree remove --force "$WT"
echo "tools/ab/libmolnextr_hip_prev.so = $(git -C "$ROOT" rev-parse --short "$REF")"
