O=gpurun_out/r03aa; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest_all.log 2>&1; tail -n 3 $O/pytest_all.log
timeout -k 10 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"
