#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "== unpipelined (CS_NO_ATTN_PP=1)"; CS_NO_ATTN_PP=1 bash tools/attn_pmc.sh gpurun_out/r06n_attn_pmc_old
echo "== pipelined"; bash tools/attn_pmc.sh gpurun_out/r06n_attn_pmc_pp
