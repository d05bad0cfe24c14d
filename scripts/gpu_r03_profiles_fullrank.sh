# re-profile of the full_rank leg only (same passes as scripts/gpu_r03_profiles.sh)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03prof; mkdir -p $O
export TMPDIR=/tmp
sed -n '/^COMMON=/,/^}/p' scripts/gpu_r03_profiles.sh > /tmp/leg_fn.sh
. /tmp/leg_fn.sh
leg c4_f32_noise1_nw8192 --noise 1.0 --walkers 8192
find $O -name "*.csv" -size +3M -delete
