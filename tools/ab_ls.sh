#!/bin/bash
bash tools/ab.sh gpurun_out/ab_ls "--steps 4 --warmup 1" raytracer_amd/variants/libmirt_lb1.so raytracer_amd/variants/libmirt_ls8.so raytracer_amd/variants/libmirt_ls6.so
echo "blocks 5 / shadow 4:"
MIRT_BLOCKS_PER_CU=5 MIRT_SHADOW_BLOCKS=4 bash tools/ab.sh gpurun_out/ab_ls5 "--steps 4 --warmup 1" raytracer_amd/variants/libmirt_ls6.so raytracer_amd/variants/libmirt_lb1.so
