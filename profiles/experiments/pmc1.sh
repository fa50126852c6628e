cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc1
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAVES" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM SQ_ACTIVE_INST_MISC SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS SQ_IFETCH SQ_VALU_MFMA_BUSY_CYCLES"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $set -d $R/gpurun_out/pmc1/$tag -o p -- python $R/profiles/experiments/exp_variant.py --systems 262144 --calls 2 > $R/gpurun_out/pmc1/$tag.log 2>&1
  python $R/profiles/pmc_dump.py $R/gpurun_out/pmc1/$tag.json hy_taylor "exp_variant --systems 262144 --calls 2 ($HEYOKA_AMD_PAIR_SPLIT)" $(find $R/gpurun_out/pmc1/$tag -name '*.db' | head -1) 2>&1 | tail -24
done
find $R/gpurun_out/pmc1 -name '*.db' -size +4M -delete
