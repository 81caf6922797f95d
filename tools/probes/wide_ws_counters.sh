#!/bin/bash
# SQ counters of k_wide_ws (256 x 256 GCN layer, weights stationary) from THE TREE'S kernel: the product build and the two measurement variants
# (PP_WS_DBG=1: no gather = the matrix stream alone; PP_WS_DBG=3: no gather, no stores).  Build the variants HERE first:
#   bash tools/probes/variant_lib.sh ws_nogather "-DPP_WS_DBG=1"; bash tools/probes/variant_lib.sh ws_matrix_only "-DPP_WS_DBG=3"
# then run ON the GPU box:  bash tools/probes/wide_ws_counters.sh  ->  gpurun_out/wide_ws_sq_counters.txt
# (counter sets in separate --pmc passes with --kernel-trace only, as MI355X_MICROARCH.md prescribes)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/wide_ws_sq_counters.txt
cp $R/pathpyg_amd/lib/libpathpyg_amd.so /tmp/keep_ws.so
{
echo "# k_wide_ws<0>, 10^7 rows x 256 x 256, De-Bruijn-shaped CSR (1.9 neighbours per row + self term), six launches per pass; built from the tree:"
echo "# product = the shipped kernel; ws_nogather = -DPP_WS_DBG=1 (no index loads, no neighbour / self rows: the matrix stream + epilogue + stores);"
echo "# ws_matrix_only = -DPP_WS_DBG=3 (additionally no stores).  rocprofv3 --pmc <set> --kernel-trace, one set per pass; tools/rocprof_pmc.py"
for T in product ws_nogather ws_matrix_only; do
  if [ $T != product ]; then cp $R/tools/probes/_bin/lib_$T.so $R/pathpyg_amd/lib/libpathpyg_amd.so; fi
  echo "== $T: time"; python $R/tools/probes/wide_ws_pmc.py 2>&1 | tail -1
  for SET in "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT"; do
    rm -rf /tmp/p_ws; rocprofv3 --pmc $SET --kernel-trace -d /tmp/p_ws -o x -- python $R/tools/probes/wide_ws_pmc.py > /dev/null 2>&1
    echo "== $T: $SET"; python $R/tools/rocprof_pmc.py $(find /tmp/p_ws -name "*.db" | head -1) --top 40 2>&1 | grep -E "k_wide_ws|^kernel" | cut -c1-170
  done
  cp /tmp/keep_ws.so $R/pathpyg_amd/lib/libpathpyg_amd.so
done
} > $O 2>&1
cp /tmp/keep_ws.so $R/pathpyg_amd/lib/libpathpyg_amd.so
head -60 $O
