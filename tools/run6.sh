cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x -k "pool_resize or general_conv or input_pipeline or fid_inception or compare_directories or conv_fast_x4 or groupnorm or dropout" 2>&1 | tail -30
