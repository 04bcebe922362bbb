for w2 in 12 8 6 5 4 3; do
  echo "w2=$w2"; EAST_HIP_REFINE_W2=$w2 EAST_HIP_TRACE=1 timeout 300 python3 bench.py --corpus zipf --docs 100 --doc-mib 1 --keyphrases 1000 --no-cpu-baseline --no-config2 --no-extras 2>&1 | python3 -c "
import sys, json
rounds=set()
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print('  build %.2f ms step %.2f' % (d['build_ms'], d['ms_per_step']))
    elif 'round' in line: rounds.add(line.strip())
for r in sorted(rounds): print('  ', r)
"
done
