for v in bl48 bl64; do bash tools/exp/kstat_lib.sh $v "bin_step" | tail -2; done
for v in fl16 fl32; do bash tools/exp/kstat_lib.sh $v "hash_fwd" | tail -3; done
