bash tools/run_evidence.sh tests 2>&1 | tail -8
bash tools/run_evidence.sh bench 2>&1 | tail -14
bash tools/run_evidence.sh profiles 2>&1 | tail -3
