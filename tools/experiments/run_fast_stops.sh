for s in 4 5 6; do echo "== stop $s"; ORB_FAST_DBGSTOP=$s bash tools/prof_latency.sh 2>&1 | grep -E "k_fast|span"; done
