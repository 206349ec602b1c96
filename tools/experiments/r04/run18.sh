#!/bin/bash
# Round 4, GPU run 18: kPf1 parity (default at 1 000 UEs, forced everywhere) + the whole GPU suite
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_run18; mkdir -p $O; cd ..
python -m pytest tests/test_gpu_round4.py -m gpu -x -q -k "per_flow" > $O/pytest_pf1.log 2>&1; echo "pytest rc $?" >> $O/pytest_pf1.log
tail -15 $O/pytest_pf1.log
python -m pytest tests -m gpu -x -q > $O/pytest_all.log 2>&1; echo "pytest rc $?" >> $O/pytest_all.log
tail -4 $O/pytest_all.log
