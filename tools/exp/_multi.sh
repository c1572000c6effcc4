for i in 1 2 3 4 5 6 7 8 9 10; do timeout 600 python -m pytest tests/test_convergence_gpu.py -x -q 2>&1 | grep -a "passed\|failed\|^E  " | tail -3; done
