cd $GRAFT_REPO_ROOT; O=gpurun_out
( python tools/bench_gn.py 256; python tools/bench_gn.py 4 bedroom ) 2>&1 | grep -v amdgpu.ids > $O/r5_gn_new.txt
( DP_HIP_LIB=$PWD/diff-pruning_amd/libdp_hip_r4norm.so python tools/bench_gn.py 256; DP_HIP_LIB=$PWD/diff-pruning_amd/libdp_hip_r4norm.so python tools/bench_gn.py 4 bedroom ) 2>&1 | grep -v amdgpu.ids > $O/r5_gn_old.txt
paste -d'|' $O/r5_gn_old.txt $O/r5_gn_new.txt | cut -c1-230
python -m pytest tests/test_kernels_gpu.py -q -x -k "groupnorm or fold" 2>&1 | tail -5
python -m pytest tests/test_e2e_gpu.py -q -k "winograd_on_every or c5_ldm or c3_bedroom or c2_cifar" --durations=8 2>&1 | tail -40
cp $O/test_report.json $O/r5_test_report_call1.json
python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/r5_bench_a.json 2> $O/r5_bench_a.err; tail -c 600 $O/r5_bench_a.json
