bash tools/run_evidence.sh tests 2>&1 | tail -30
