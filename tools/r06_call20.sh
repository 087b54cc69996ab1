cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1700 python -m pytest tests -m gpu -q 2>&1 | tail -5
