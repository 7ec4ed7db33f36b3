cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for nb in 64 14 2; do
SM_BATCH=$nb timeout 600 python tools/gemm_tok_smallm.py > gpurun_out/r04_i_tok_smallm_b$nb.txt 2>&1
done
cut -c1-400 gpurun_out/r04_i_tok_smallm_b64.txt
