#!/bin/bash
# diagnostic build (-DVS_STAMPS: s_memrealtime stamps by thread 0 of the workgroups) of the library into scratch/stamps/
# (git-ignored; never the product library).  The stamp readers in this directory load it through pkg.LIB_PATH.
set -e
cd /root/repo/hai-25-rag-on-edge_amd/csrc
O=/root/repo/scratch/stamps; mkdir -p $O
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-function -DVS_STAMPS"
for f in vs_scan vs_seed_merge vs_ivf vs_build vs_scan_one vs_api vs_q8; do /opt/rocm/bin/hipcc $F -c $f.hip -o $O/$f.o & done
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden -x c++ -c vs_host.cpp -o $O/vs_host.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $O/libvsearch_hip.so $O/*.o -lpthread
ls -la $O/libvsearch_hip.so
