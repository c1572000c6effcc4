for v in bins256 bins256t256 t256; do bash tools/exp/kstat_lib.sh $v "hash_b" | tail -4; done
