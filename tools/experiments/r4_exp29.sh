#!/bin/bash
# round 4, batch 29: PMC instruction mix of both flash attentions after the packed fp32 left them (compare profiles/r03_pmc_attention_summary.txt)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4e29
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_LDS_IDX_ACTIVE"; do
  timeout -k 10 200 rocprofv3 --pmc $set -d $O/pmc_attn_x3 --output-format csv -- python3 $R/tools/prof_x3.py > /dev/null 2>> $O/pmc_attn_x3.err || exit 1
  timeout -k 10 200 rocprofv3 --pmc $set -d $O/pmc_attn_bf16 --output-format csv -- python3 $R/tools/prof_attn.py > /dev/null 2>> $O/pmc_attn_bf16.err || exit 1
done
python3 $R/tools/pmc_summary.py $O/pmc_attn_x3 > $O/pmc_attn_x3_summary.txt
python3 $R/tools/pmc_summary.py $O/pmc_attn_bf16 > $O/pmc_attn_bf16_summary.txt
grep -A 17 "attn_" $O/pmc_attn_x3_summary.txt $O/pmc_attn_bf16_summary.txt | cut -c1-140
