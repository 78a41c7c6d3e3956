cd $GRAFT_REPO_ROOT; O=gpurun_out
for t in 384 512; do DP_WINO_MIN_TILES=$t python bench.py --config ldm --steps 2 --warmup 1 --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ldm WINO_MIN_TILES=$t', round(d['ms_per_step'],2))"; done
for t in 384 512; do DP_WINO_MIN_TILES=$t python bench.py --config bedroom256 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bedroom WINO_MIN_TILES=$t', round(d['ms_per_step'],2))"; done
for t in 384 512; do DP_WINO_MIN_TILES=$t python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cifar256 WINO_MIN_TILES=$t', round(d['ms_per_step'],2))"; done
