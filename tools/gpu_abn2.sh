#!/bin/bash
# GPU box, repo root: several (workload, precision) pairs for two or more builds of the library.
# usage: tools/gpu_abn2.sh "c2:f64 c4:f64" <lib> [<lib> ...]
pairs=$1; shift
for wp in $pairs; do
  echo "== $wp"
  tools/gpu_abn.sh ${wp%%:*} ${wp##*:} "$@" || exit 1
done
