bash tools/run_evidence.sh pmc 2>&1 | tail -8
ls -la gpurun_out/round5_pmc*
