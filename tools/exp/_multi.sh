for v in b100 b110 b125 wide3; do bash tools/exp/kstat_lib.sh $v "wgrad" | tail -2; done
