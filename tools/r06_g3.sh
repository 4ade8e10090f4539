bash tools/record_evidence.sh 2>&1 | tail -4
