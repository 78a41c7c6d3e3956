cd $GRAFT_REPO_ROOT; O=gpurun_out
python tools/profile_shapes.py --config pruned --batch 128 2>&1 | grep -v amdgpu.ids > $O/r5_shapes_pruned128.txt
python tools/profile_shapes.py --config pruned --batch 256 --forward-only 2>&1 | grep -v amdgpu.ids > $O/r5_shapes_pruned256_fwd.txt
python tools/profile_shapes.py --config cifar --batch 256 2>&1 | grep -v amdgpu.ids > $O/r5_shapes_cifar256.txt
python tools/profile_shapes.py --config ldm --batch 12 --forward-only 2>&1 | grep -v amdgpu.ids > $O/r5_shapes_ldm_fwd12.txt
head -50 $O/r5_shapes_pruned128.txt
