O=gpurun_out/r06q; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/suite.log 2>&1; grep -n "passed\|failed" $O/suite.log | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python tools/soak.py --help 2>&1 | head -20
