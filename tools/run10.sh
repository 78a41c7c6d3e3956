cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x 2>&1 | tail -5
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; b=json.load(sys.stdin); print(b['ms_per_step'], b['value'], b['config']['kernel_launches_per_step'], b['config']['tail_ms'])"
