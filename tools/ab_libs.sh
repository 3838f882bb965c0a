#!/bin/bash
# Runs a tool once per build of the library (multigrid_amd/libmg_hip*.so variants), for A/B comparisons.
# usage: bash tools/ab_libs.sh "<variant names, '' = the default build>" <command...>
R=${GRAFT_REPO_ROOT:-$PWD}
variants=$1; shift
for v in $variants; do
  lib=$R/multigrid_amd/libmg_hip_$v.so
  [ "$v" = base ] && lib=$R/multigrid_amd/libmg_hip.so
  echo "=== $v"
  MG_HIP_LIB=$lib timeout -k 10 200 "$@" || exit 1
done
