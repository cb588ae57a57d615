set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
P=gpurun_out/prof_gn; rm -rf $P; mkdir -p $P gpurun_out/profiles_r04
rocprofv3 --kernel-trace --stats --output-format csv -d $P/st -o st -- python3 tools/probes/gn_bench.py > $P/st.log 2>&1
cp $(find $P/st -name "*kernel_stats.csv" | head -1) gpurun_out/profiles_r04/r04_groupnorm_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $P/f -o f -- python3 tools/probes/gn_bench.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $P/w -o w -- python3 tools/probes/gn_bench.py > /dev/null 2>&1
python3 tools/probes/pmc_summary.py gpurun_out/profiles_r04/r04_groupnorm_pmc_raw.json fetch=$(find $P/f -name "*counter_collection.csv" | head -1) write=$(find $P/w -name "*counter_collection.csv" | head -1)
python3 tools/probes/pmc_derive.py gpurun_out/profiles_r04/r04_groupnorm_pmc_raw.json gpurun_out/profiles_r04/r04_groupnorm_pmc.json images=64 note="tools/probes/gn_bench.py: GroupNorm(32)+SiLU forward (stats + apply), recompute, backward (reduce + apply, with and without the residual) of a 192-channel tensor at 256x256 and 128x128, 64 images; bytes per launch are AVERAGES over both sizes and the launches of the timing loops"
grep -v amdgpu $P/st.log | tail -12
rm -rf $P
