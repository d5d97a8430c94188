#!/bin/bash
# Host-side AddressSanitizer + UBSan over the library's C++ host code on a GPU: the C client and the command line,
# both linked against a host-instrumented build of the library.  The device code is not instrumented.  Build first, here:
#   tools/build_variant.sh san -Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined -g
#   CL=/opt/rocm/lib/llvm/bin/clang   (the clients need the same sanitizer runtime as the library)
#   $CL -std=c99 -g -O1 -fsanitize=address,undefined tests/c_abi/c_abi_check.c -o build_variants/c_abi_check_san \
#       -Lbuild_variants -lrrt_san -lm -Wl,-rpath,'$ORIGIN'
#   ${CL}++ -O1 -g -std=c++17 -fsanitize=address,undefined rustraytracer_amd/host/rrt_main.cpp -o build_variants/rrt_san \
#       -pthread -Lbuild_variants -lrrt_san -lz -Wl,-rpath,'$ORIGIN'
# then: gpurun -- 'tools/san_gpu.sh'.  Last run: clean (C client, Cornell box, smallpt, Cornell + cube through the
# trees and the wavefront pipeline in both precisions, the million-triangle build).
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1
B=build_variants
timeout -k 10 120 $B/c_abi_check_san gpu 2>&1 | tail -15 || exit 1
for cfg in cornell_box/render_low_quality smallpt/render_low_quality; do
  scene=configs/$(dirname $cfg)/scene.toml
  timeout -k 10 120 $B/rrt_san -s $scene -r configs/$cfg.toml -o gpurun_out/san_$(dirname $cfg).png --stats-json gpurun_out/san_$(dirname $cfg).json 2>&1 | tail -8 || exit 1
done
timeout -k 10 120 $B/rrt_san -s configs/cornell_cube/scene.toml -r configs/cornell_cube/render_c4.toml -o gpurun_out/san_cube.png --accel lbvh --pipeline wavefront 2>&1 | tail -8 || exit 1
# the tree built on the GPU, next-event estimation through the shadow queue, the f64 queues, the tracked closest hit
timeout -k 10 120 $B/rrt_san -s configs/cornell_cube/scene.toml -r configs/cornell_box/render_low_quality.toml -o gpurun_out/san_cube2.png --accel sah_gpu --pipeline wavefront --next-event 2>&1 | tail -8 || exit 1
timeout -k 10 120 $B/rrt_san -s configs/cornell_cube/scene.toml -r configs/cornell_box/render_low_quality.toml -o gpurun_out/san_cube3.png --accel sah_gpu --pipeline wavefront --precision f64 --next-event 2>&1 | tail -8 || exit 1
timeout -k 10 120 $B/rrt_san -s configs/cornell_cube/scene.toml -r configs/cornell_box/render_low_quality.toml -o gpurun_out/san_cube4.png --accel sah_gpu --pipeline fused --exact-hits 2>&1 | tail -8 || exit 1
# a million triangles: the threaded host preparation and the chunked build kernels (a small image: the build is what is checked)
if [ -f configs/mesh_1m/scene.toml ] && [ -f configs/cornell_box/render_low_quality.toml ]; then
  python3 tools/make_c5_assets.py > /dev/null || exit 1
  timeout -k 10 300 $B/rrt_san -s configs/mesh_1m/scene.toml -r configs/cornell_box/render_low_quality.toml -o gpurun_out/san_mesh.png --stats-json gpurun_out/san_mesh.json 2>&1 | tail -8 || exit 1
fi
echo "san_gpu ok"
