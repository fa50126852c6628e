cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export HEYOKA_AMD_WAVE_ROLES=1
mkdir -p $R/gpurun_out/pmc2
DBS=""
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM SQ_WAVES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set -d $R/gpurun_out/pmc2/s$i -o p -- python $R/profiles/experiments/exp_variant.py --systems 262144 --calls 2 --dt 40 > $R/gpurun_out/pmc2/s$i.log 2>&1
  DBS="$DBS $(find $R/gpurun_out/pmc2/s$i -name '*.db' | head -1)"
done
python $R/profiles/pmc_dump.py $R/gpurun_out/pmc2/r02_outer_ss_v4_wave_roles_sq_counters.json hy_taylor "cluster v4 (wave roles, opt-in HEYOKA_AMD_WAVE_ROLES=1): exp_variant --systems 262144 --calls 2 --dt 40, last dispatch" $DBS | tail -30
find $R/gpurun_out/pmc2 -name '*.db' -size +4M -delete
