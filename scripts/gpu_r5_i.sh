#!/bin/bash
# staged value-window pooling: variant libraries named on the command line (EPA_VARIANT builds)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5i; mkdir -p $O; : > $O/pool_value_ab.txt
for v in "$@"; do
  echo "== $v" >> $O/pool_value_ab.txt
  L=$PWD/echopype_amd/lib/libechopype_amd_$v.so; [ "$v" = base ] && L=$PWD/echopype_amd/lib/libechopype_amd.so
  ECHOPYPE_AMD_LIB=$L timeout 600 python scripts/perf_pool_value.py ${PV_ARGS:-} >> $O/pool_value_ab.txt 2>&1
done
cat $O/pool_value_ab.txt
