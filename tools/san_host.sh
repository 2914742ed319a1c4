#!/bin/bash
# Host BVH builders (reference tree, 4-wide fold, own SAH tree; threaded) under ASan+UBSan and under TSan.  CPU only.
set -e
cd "$(dirname "$0")/../raytracer_amd/csrc"
mkdir -p /tmp/san
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -I . -o /tmp/san/drv ../../tools/san_host_bvh.cpp host_bvh.cpp -lpthread
/tmp/san/drv
g++ -std=c++17 -O1 -g -fsanitize=thread -I . -o /tmp/san/drv_t ../../tools/san_host_bvh.cpp host_bvh.cpp -lpthread
MIRT_BUILD_THREADS=8 /tmp/san/drv_t
echo "sanitizers: clean"
