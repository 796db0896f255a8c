run() { python tools/batch_bench.py "$@" 2>/dev/null | sed -E 's/tile=0 //; s/ tail=1//; s/ qb=0 groups=0//; s/, [0-9.e+]+ element.*//'; }
for cube in "1000 5 1000" "2000 5 2000" "4000 5 4000" "10000 5 10000"; do for b in 16 32 48 64 96 128 192 256; do for m in 1 257; do run $cube f64 $b 0 $m 5; done; done; done
