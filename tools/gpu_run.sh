#!/bin/bash
# One parametrised script for this project's `gpurun` calls (replaces the per-call
# scripts of round 2).  Runs on the GPU box from the repo root; everything it keeps
# goes under gpurun_out/<tag>/.
#
#   tools/gpu_run.sh <tag> <step> [<step> ...]
#
# steps:
#   tests[:pytest -k expression]   pytest -m gpu (-x), log to pytest.log
#   testfile:<path>[:k-expr]       one test file
#   bench[:extra args]             python bench.py --steps 20 --warmup 5 (driver's shape) -> bench.json
#   benchN:<n>[:extra args]        plain `python bench.py --gpus n` (self-launch; gloo + shared GPU rehearsal)
#   prof[:tag[:extra args]]        rocprofv3 --kernel-trace --stats of bench.py (no cpu baseline) -> prof[_tag]/
#   pmc:<workload>[:num]           tools/pmc_collect.sh passes for a workload -> pmc_<tag>_<workload>/
#   probe:<mode>:<workload>[:field[:lib[:uncached]]]   tools/sustained_probe.py (steady-state kernel time)
#   py:<script and args>           python <script...> > <tag>/<script>.out
#   lib:<variant.so>               export ROX_LIB for the following steps ("lib:" resets)
#   env:NAME=VALUE                 export for the following steps ("env:NAME=" unsets)
#   smoke                          __graft_entry__.smoke()
set -u
cd "${GRAFT_REPO_ROOT:-$PWD}"
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export ROX_TEST_RECORDS="$PWD/$OUT/test_records.jsonl"
for step in "$@"; do
  kind=${step%%:*}
  rest=""; [ "$step" != "$kind" ] && rest=${step#*:}
  echo "=== $step"
  case $kind in
    tests)
      if [ -n "$rest" ]; then timeout 2400 python -m pytest tests -m gpu -x -q -k "$rest" > "$OUT/pytest.log" 2>&1
      else timeout 2400 python -m pytest tests -m gpu -x -q > "$OUT/pytest.log" 2>&1; fi
      echo "rc=$?" >> "$OUT/pytest.log"; tail -8 "$OUT/pytest.log" ;;
    testfile)
      f=${rest%%:*}; k=""; [ "$rest" != "$f" ] && k=${rest#*:}
      log="$OUT/pytest_$(basename "$f" .py).log"
      if [ -n "$k" ]; then timeout 2400 python -m pytest "$f" -m gpu -x -q -k "$k" > "$log" 2>&1
      else timeout 2400 python -m pytest "$f" -m gpu -x -q > "$log" 2>&1; fi
      echo "rc=$?" >> "$log"; tail -12 "$log" ;;
    bench)
      timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 $rest > "$OUT/bench.json" 2> "$OUT/bench.err"
      echo "rc=$? lines=$(wc -l < "$OUT/bench.json")"; tail -3 "$OUT/bench.err"
      python - "$OUT/bench.json" <<'PY'
import json, sys
b = json.load(open(sys.argv[1]))
print('value %.4g  ms/step %.4f  cold %.4f  frac %.3f  hits_ms %.4f  spot_ms %.3f' % (
    b['value'], b['ms_per_step'], b['cold_ms_per_step'], b['roofline']['frac'],
    b['roofline_hits']['kernel_ms'], b['spot_diagram']['wallclock_ms']))
for k, c in (b.get('configs') or {}).items():
    if isinstance(c, dict) and 'hits' in c:
        loop = lambda d: (' (per-grid launches %.3f)' % d['kernel_ms_per_pass_one_launch_per_grid']) if 'kernel_ms_per_pass_one_launch_per_grid' in d else ''
        print(k, 'hits %.3f ms%s' % (c['hits']['kernel_ms_per_pass'], loop(c['hits'])),
              ('full %.3f ms%s %.3f of 8 TB/s' % (c['full']['kernel_ms_per_pass'], loop(c['full']), c['full']['frac_of_8000'])) if 'full' in c else '')
    else:
        print(k, c)
s = b.get('strong_scaling') or {}
for p in ('c5', 'c4'):
    for ex in ('rccl', 'host'):
        print(p, ex, (s.get(p) or {}).get(ex) if s.get(p) else s)
PY
      ;;
    benchN)
      n=${rest%%:*}; extra=""; [ "$rest" != "$n" ] && extra=${rest#*:}
      ROX_BENCH_BACKEND=gloo ROX_BENCH_SHARE_GPU=1 timeout 1500 python bench.py --gpus $n --steps 20 --warmup 5 $extra > "$OUT/bench_n$n.json" 2> "$OUT/bench_n$n.err"
      echo "n=$n rc=$? lines=$(wc -l < "$OUT/bench_n$n.json")"; tail -5 "$OUT/bench_n$n.err"
      python - "$OUT/bench_n$n.json" <<'PY'
import json, sys
b = json.load(open(sys.argv[1]))
print(b['n_gpus'], b['ranks_seen_by_backend'], b['value'], b['ms_per_step'])
s = b['strong_scaling']
for p in ('c5', 'c4'):
    for ex in ('rccl', 'host'):
        print(p, ex, s[p].get(ex))
PY
      ;;
    prof)     # prof[:tag[:extra bench args]]  e.g. prof:main:--no-configs --no-strong
      R=$PWD
      ptag=${rest%%:*}; pextra=""; [ "$rest" != "$ptag" ] && pextra=${rest#*:}
      pdir="prof${ptag:+_$ptag}"
      (cd /tmp; export TMPDIR=/tmp; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/$pdir" -o bench -- python "$R/bench.py" --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline $pextra > "$R/$OUT/${pdir}_bench.json" 2> "$R/$OUT/$pdir.err")
      echo "rc=$?"; find "$OUT/$pdir" -name "*kernel_stats.csv" | head -1 | xargs -r head -12 | cut -c1-220
      find "$OUT/$pdir" -name "*kernel_trace.csv" -size +8M -delete ;;
    pmc)
      wl=${rest%%:*}
      bash tools/pmc_collect.sh "${TAG}_$wl" $(echo "$rest" | tr ':' ' ') ;;
    probe)
      IFS=: read -r mode wl field lib unc <<< "$rest"
      [ -n "${lib:-}" ] && export ROX_LIB="$PWD/$lib"
      timeout 300 python tools/sustained_probe.py --mode "$mode" --workload "${wl:-dblgauss_c2}" --field "${field:-0}" --seconds 2 ${unc:+--uncached} 2>/dev/null \
        | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$rest', d['lib'], 'mean_us', round(d['mean_us'],1), 'last_quarter', round(d['last_quarter_mean_us'],1))" | tee -a "$OUT/probe.txt"
      unset ROX_LIB ;;
    py)
      name=$(echo "$rest" | awk '{print $1}' | xargs basename)
      timeout 1800 python $rest >> "$OUT/$name.out" 2> "$OUT/$name.err"; echo "rc=$?"; tail -1 "$OUT/$name.out" | cut -c1-1500; tail -3 "$OUT/$name.err" ;;
    lib)      # ROX_LIB for the following steps ("lib:" alone: back to the product library)
      if [ -n "$rest" ]; then export ROX_LIB="$PWD/$rest"; else unset ROX_LIB; fi
      echo "ROX_LIB=${ROX_LIB:-}" ;;
    env)      # env:NAME=VALUE exports, env:NAME= unsets, for the following steps
      if [ -n "${rest#*=}" ]; then export "$rest"; else unset "${rest%%=*}"; fi; echo "$rest" ;;
    smoke)
      timeout 600 python __graft_entry__.py smoke 2>&1 | tail -3 ;;
    *) echo "unknown step $step" ;;
  esac
done
