#!/bin/bash
# A/B of an environment setting against the defaults on the shares and on every workload: tools/ab_env.sh "VAR=x VAR2=y"
echo "== defaults"; bash tools/share.sh; bash tools/workloads_full.sh
echo "== $1"; env $1 bash tools/share.sh; env $1 bash tools/workloads_full.sh
