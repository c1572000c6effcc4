for v in wnt lnt wnt lnt; do bash tools/ab_lib.sh $v 2; done
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -1
