#!/bin/bash
# round 6, first call: per-kernel us at three ray counts (fixed vs per-ray cost), and the gather with / without the XCD-affine level schedule
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06
for rays in 512 1024 4096; do
  echo "== --rays $rays" >> $R/gpurun_out/r06/kstat_rays.txt
  bash $R/tools/exp/kstat_args.sh --rays $rays >> $R/gpurun_out/r06/kstat_rays.txt 2>&1
done
bash $R/tools/exp/kstat_env.sh HOLOSCENE_HASH_SCHEDULE 1 0 'k_hash|k_sdf_mlp2|k_rr_fwd' > $R/gpurun_out/r06/gather_schedule.txt 2>&1
