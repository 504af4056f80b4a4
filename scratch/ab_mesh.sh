# usage (GPU box): bash scratch/ab_mesh.sh base v1 ...   -- scratch/split_probe.py (1M mesh, 1 / 16 / 64 spp) for the product library and each build/var_<v>
for v in "$@"; do
  if [ $v = base ]; then unset PT_LIBDIR; else export PT_LIBDIR=$PWD/rust_pathtracer_amd/build/var_$v; fi
  echo "== $v"; timeout -k 10 200 python -u scratch/split_probe.py "" 2>&1 | grep -v amdgpu.ids || exit 1
done
