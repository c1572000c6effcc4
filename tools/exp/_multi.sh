bash tools/ab_lib.sh tnt 2 --objects 64
bash tools/ab_lib.sh tnt 2
