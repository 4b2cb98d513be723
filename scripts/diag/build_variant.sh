#!/bin/bash
# scratch/build_variant.sh <name> <extra -D flags...>: a variant of the library into scratch/libs/lib_<name>.so
set -e
name=$1; shift
cd /root/repo/hai-25-rag-on-edge_amd/csrc
mkdir -p /root/repo/scratch/libs; O=/root/repo/scratch/libs/obj_$name; mkdir -p $O
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-function $*"
for f in vs_scan vs_seed_merge vs_ivf vs_build vs_scan_one vs_api vs_q8; do /opt/rocm/bin/hipcc $F -c $f.hip -o $O/$f.o & done
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden -x c++ -c vs_host.cpp -o $O/vs_host.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o /root/repo/scratch/libs/lib_$name.so $O/*.o -lpthread
rm -rf $O
ls -la /root/repo/scratch/libs/lib_$name.so
