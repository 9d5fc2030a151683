for rep in 1 2; do
for lib in base trim1; do
  for wl in C2 C4 C5; do
    echo "== $lib $wl default"; LK_ENGINE_LIB=$PWD/build/tune/liblk_$lib.so timeout -k 10 200 python3 scripts/quick_solve.py $wl 10 2>&1 | tail -1
  done
  echo "== $lib C2 ref"; LK_REF_ORDER=1 LK_ENGINE_LIB=$PWD/build/tune/liblk_$lib.so timeout -k 10 200 python3 scripts/quick_solve.py C2 10 2>&1 | tail -1
done
done
