# usage (GPU box): bash scratch/ab_variants.sh base v1 v2 ...   -- scratch/ab_two.py for the product library and each build/var_<v>
for v in "$@"; do
  if [ $v = base ]; then unset PT_LIBDIR; else export PT_LIBDIR=$PWD/rust_pathtracer_amd/build/var_$v; fi
  r=$(timeout -k 10 120 python scratch/ab_two.py 2>/dev/null) || exit 1
  echo "$v: $r"
done
