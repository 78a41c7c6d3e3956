cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x -k "stride2_dgrad or finetune" 2>&1 | tail -2
for cfg in "A=1" "DP_NO_S2_PARITY=1" "A=2" "DP_NO_S2_PARITY=2"; do
  echo "== $cfg"
  env $cfg python tools/bench_secondary.py 2>&1 | grep -v amdgpu | grep -i "finetune"
done
