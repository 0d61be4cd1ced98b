#!/bin/bash
O=gpurun_out/ab2; mkdir -p $O; : > $O/log4.txt
timeout 2400 python -m pytest tests -m gpu -x -q -s 2>&1 | grep -a "emulated\|passed\|failed\|Error" >> $O/log4.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 >> $O/log4.txt
cat $O/log4.txt
