# usage: scratch/ablate.sh name "flags" ...   builds rust_pathtracer_amd/build/ablate_<name>/
set -e
cd /root/repo/rust_pathtracer_amd
while [ $# -gt 0 ]; do
  name=$1; flags=$2; shift 2
  d=build/ablate_$name; rm -rf $d; mkdir -p $d
  for f in raygen trace shade path resolve ptgpu; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $flags -c csrc/device/$f.hip -o $d/$f.o &
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $d/libptgpu.so $d/*.o
  cp lib/libpthost.so $d/
done
