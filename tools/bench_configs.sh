#!/bin/bash
# The other BASELINE configurations next to the default bench line (run on the GPU box through gpurun):
# one JSON line per configuration into gpurun_out/configs.jsonl, copied to profiles/ afterwards.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/configs.jsonl
mkdir -p gpurun_out; : > $OUT
run() { timeout 300 python3 bench.py --full-line --no-cpu-baseline "$@" 2>/dev/null | tail -1 >> $OUT; }
run                                                           # configs[1]: 1 x 64 MiB, text mode, 1 000 keyphrases
run --mode direct                                             # configs[1], get_ast([one 64 Mi string])
run --docs 256 --doc-mib 1 --keyphrases 10000                 # configs[2]: 256 x 1 MiB, 10 000 keyphrases
run --docs 256 --doc-mib 1 --keyphrases 10000 --denormalized  # the CLI's -d
run --corpus zipf --docs 100 --doc-mib 1 --keyphrases 1000    # config 5 stand-in (Zipf word stream)
run --docs 16 --doc-mib 16 --keyphrases 1000
python3 - <<'PY'
import json
for line in open("gpurun_out/configs.jsonl"):
    d = json.loads(line)
    print("%-88s step %7.2f ms  build %6.2f  score %5.2f  %.2e chars/s  %.2e scores/s" % (
        d["config"]["workload"][:88], d["ms_per_step"], d["build_ms"], d["score_ms"], d["value"], d["keyphrase_scores_per_s"]))
PY
timeout 600 python3 tools/natural_text_bench.py --resample-mib 64 --doc-mib 1 2>&1 | grep -E "^build|^symbols" > gpurun_out/natural_text.txt
timeout 300 python3 tools/time_config1.py 2>&1 | tail -1 >> gpurun_out/natural_text.txt
cat gpurun_out/natural_text.txt
