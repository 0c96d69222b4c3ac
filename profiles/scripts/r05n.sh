#!/bin/bash
# round 5: the whole GPU suite once more on the final tree (as the driver runs it: -x)
O=gpurun_out/r05n
mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4
