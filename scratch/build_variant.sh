# usage: scratch/build_variant.sh name "flags"   -> rust_pathtracer_amd/build/var_<name>/{libptgpu.so,libpthost.so}; run with PT_LIBDIR=...
set -e
cd /root/repo/rust_pathtracer_amd
name=$1; flags=$2
d=build/var_$name; rm -rf $d; mkdir -p $d
for f in raygen trace shade path resolve guide_build ptgpu; do
  # (guide_build.hip pulls in rocPRIM: slow to compile, and no variant flag reaches it)
  if [ $f = guide_build ] && [ -f build/guide_build.o ]; then cp build/guide_build.o $d/; continue; fi
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $flags -c csrc/device/$f.hip -o $d/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $d/libptgpu.so $d/*.o -ldl
cp lib/libpthost.so $d/
rm -f $d/*.o
