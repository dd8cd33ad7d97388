cd $GRAFT_REPO_ROOT
timeout 100 python tools/sustained_probe.py --mode hits --seconds 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('hits', round(d['mean_us'],1), round(d['last_quarter_mean_us'],1))"
timeout 100 python tools/sustained_probe.py --mode full --seconds 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('full', round(d['mean_us'],1), round(d['last_quarter_mean_us'],1))"
timeout 200 python tools/ab_bench.py --check --reps 3 2>/dev/null
