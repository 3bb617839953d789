#!/bin/bash
# tools/.ab/build_variant.sh NAME "-DFLAGS" file1.hip file2.hip ... : variant objects for the listed files, everything else from build/
NAME=$1; FLAGS=$2; shift 2
cd /root/repo/generative-turbulence_amd/csrc
mkdir -p build_$NAME
EXCL=""
for f in "$@"; do
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-slp-vectorize $FLAGS -c $f -o build_$NAME/${f%.hip}.o ) &
  EXCL="$EXCL build/${f%.hip}.o"
done
wait
OBJS=$(ls build/*.o | grep -v -F -f <(echo $EXCL | tr ' ' '\n'))
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/tools/.ab/libtdx_$NAME.so $OBJS build_$NAME/*.o
ls -la /root/repo/tools/.ab/libtdx_$NAME.so
