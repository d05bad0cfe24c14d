# re-profile of the headline leg only (same passes as scripts/gpu_r03_profiles.sh)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03prof; mkdir -p $O
export TMPDIR=/tmp
sed -n '/^COMMON=/,/^}/p' scripts/gpu_r03_profiles.sh > /tmp/leg_fn.sh
. /tmp/leg_fn.sh
leg c4_f32_noise0.1_nw49152
find $O -name "*.csv" -size +3M -delete
