for w in 12 9 8 7 6; do echo "== ROUND_W2=$w"; EAST_HIP_ROUND_W2=$w bash tools/zipf_quick.sh 2>&1 | head -2 | cut -c1-420; done
