#!/bin/bash
# A/B of the write-through (sc1) bulk stores on ONE box: STGCN_WT_STORES = 0 (plain), 1 (epilogue stores), 2 (also the in-loop stores)
OUT=$1
for L in 0 1 2; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DSTGCN_BACKEND_NAME=\"hip-gfx950\" -DSTGCN_WT_STORES=$L stgcn_amd/csrc/stgcn_capi.hip -o /tmp/libstgcn_wt$L.so 2>/dev/null
done
for rep in 1 2; do for L in 0 1 2; do
  echo "== STGCN_WT_STORES=$L (run $rep)"
  STGCN_AMD_LIB=/tmp/libstgcn_wt$L.so python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-gpu-baseline | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['per_kernel_us_per_step']
print(d['ms_per_step'], {n: k[n] for n in ('tc1_bwd@1','tc2_ln_fwd@0','tc2_ln_fwd@1','tconv_fwd.tc1@1','tc2_bwd@0','gconv_fwd@0','head.fc_fwd@0','head.tconv_bwd_data@0','head.tconv_fwd@0')})"
done; done | tee $OUT/wt_ab.txt
