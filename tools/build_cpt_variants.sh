#!/bin/bash
# usage: tools/build_cpt_variants.sh "<VARIANT> <STGN> <P2> <SKIPW> <ENDBAR> [extra -D flags]" ...   -- builds tools/cpt_one_<tag> for each spec (A/B harness, RCX_STAMPS)
cd "$(dirname "$0")/.."
for spec in "$@"; do
  set -- $spec
  v=$1; st=$2; p2=$3; sk=$4; eb=$5; shift 5
  tag="v${v}_s${st}p${p2}k${sk}e${eb}$(echo "$*" | tr -d ' =-' | tr -c 'A-Za-z0-9_\n' '_')"
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize -Xclang -target-feature -Xclang -load-store-opt -DRCX_STAMPS \
     -DVARIANT=$v -DSTGN=$st -DRCX_CPT_STG_P2=$p2 -DRCX_CPT_SKIPW=$sk -DRCX_CPT_ENDBAR=$eb "$@" tools/cpt_one.hip -o tools/cpt_one_$tag 2>&1 | grep -E "error" -A5
  echo "built tools/cpt_one_$tag"
done
