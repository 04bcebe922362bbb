for sb in 26 34 42 48; do
  EAST_HIP_HT_SB=$sb EAST_PROFILE=1 timeout 300 python3 tools/natural_text_bench.py --resample-mib 64 2>/dev/null | grep -E "^build|launches" | head -8 | sed "s/^/sb$sb /"
done
