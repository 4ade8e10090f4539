cd $GRAFT_REPO_ROOT
for i in $(seq 1 14); do timeout 300 python -m pytest tests/test_hip_train.py -x -q -s -m gpu -k "fused_sgd_keeps" 2>&1 | grep -E "^E  |passed|failed|worst" | head -3; done
