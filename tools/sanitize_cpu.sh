#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer over everything of this tree that runs on the CPU (there is no GPU sanitizer on
# this pool): the oracle (plain C / C++) and the library's host-only sources (Delaunay, the MSA tree builder, the ELAS support
# point filters), driven by the CPU test suite.
#
#     tools/sanitize_cpu.sh            # prints the tail of both runs and the number of sanitizer reports (expected: 0)
#
# The instrumented libraries are built in a temporary directory and swapped in for the duration of the run only; the originals
# are put back on exit.  `alloc_dealloc_mismatch=0`: the reference's own libelas (oracle/_ref, when it is there) frees a
# `new[]` array with `free` (elas.cpp:1509,1559) - not this tree's code.  tests/test_ref_elas.py is left out of the oracle
# run for a related reason: libelas reads malloc'ed memory it never wrote, so under ASan's fill pattern its Middlebury-setting
# maps differ from the committed vectors.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d /tmp/svo_san.XXXXXX)
PKG="$ROOT/stereo-semantic-vo_amd"
GCC_ASAN=$(gcc -print-file-name=libasan.so)
GCC_UBSAN=$(gcc -print-file-name=libubsan.so)
CLANG_ASAN=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
restore() {
  [ -f "$T/libsvo_oracle.orig.so" ] && cp -p "$T/libsvo_oracle.orig.so" "$ROOT/oracle/libsvo_oracle.so"
  [ -f "$T/libsvo_hip.orig.so" ] && cp -p "$T/libsvo_hip.orig.so" "$PKG/libsvo_hip.so"
  rm -rf "$T"
}
trap restore EXIT
make -C "$PKG" -j8 -s && make -C "$ROOT/oracle" -s || exit 1

# 1. the oracle
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer -g -O1 -fPIC -ffp-contract=off -fno-fast-math"
for f in "$ROOT"/oracle/orc_*.c; do gcc $SAN -std=c11 -D_GNU_SOURCE -c "$f" -o "$T/$(basename "$f" .c).o" || exit 1; done
g++ $SAN -std=c++17 -c "$ROOT/oracle/orc_msa_graph.cpp" -o "$T/orc_msa_graph.o" || exit 1
g++ -shared -fsanitize=address,undefined -o "$T/libsvo_oracle.so" "$T"/orc_*.o -lm || exit 1
cp -p "$ROOT/oracle/libsvo_oracle.so" "$T/libsvo_oracle.orig.so"
cp "$T/libsvo_oracle.so" "$ROOT/oracle/libsvo_oracle.so"
(cd "$ROOT" && LD_PRELOAD="$GCC_ASAN $GCC_UBSAN" ASAN_OPTIONS=detect_leaks=0:alloc_dealloc_mismatch=0 \
  python -m pytest tests -q -m "not gpu" -p no:cacheprovider --deselect tests/test_ref_elas.py > "$T/oracle.log" 2>&1)
cp -p "$T/libsvo_oracle.orig.so" "$ROOT/oracle/libsvo_oracle.so"
echo "== oracle under ASan + UBSan"; tail -2 "$T/oracle.log"
N1=$(grep -c "runtime error\|ERROR: AddressSanitizer" "$T/oracle.log")

# 2. the library's host-only sources (the device objects are linked as they are)
HSAN="-fsanitize=address,undefined -fno-gpu-sanitize -shared-libsan -fno-omit-frame-pointer -g -O1 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math"
for f in svo_delaunay svo_msa_graph; do
  /opt/rocm/bin/hipcc $HSAN --offload-arch=gfx950 -c "$PKG/csrc/$f.hip" -o "$T/$f.o" || exit 1
done
/opt/rocm/bin/hipcc $HSAN -c "$PKG/csrc/svo_elas_filter.cc" -o "$T/svo_elas_filter.o" || exit 1
OBJS=""
for f in svo_api svo_orb svo_stereo svo_match svo_pose svo_fmat svo_track svo_elas svo_msa; do OBJS="$OBJS $PKG/csrc/$f.o"; done
/opt/rocm/bin/hipcc -shared --offload-arch=gfx950 -pthread -fsanitize=address,undefined -shared-libsan -o "$T/libsvo_hip.so" \
  $OBJS "$T/svo_delaunay.o" "$T/svo_msa_graph.o" "$T/svo_elas_filter.o" || exit 1
cp -p "$PKG/libsvo_hip.so" "$T/libsvo_hip.orig.so"
cp "$T/libsvo_hip.so" "$PKG/libsvo_hip.so"
(cd "$ROOT" && LD_PRELOAD="$CLANG_ASAN" ASAN_OPTIONS=detect_leaks=0:alloc_dealloc_mismatch=0 \
  python -m pytest tests/test_elas_delaunay.py tests/test_elas_filter.py tests/test_msa.py tests/test_abi.py -q -m "not gpu" \
  -p no:cacheprovider > "$T/host.log" 2>&1)
cp -p "$T/libsvo_hip.orig.so" "$PKG/libsvo_hip.so"
echo "== library host sources under ASan + UBSan"; tail -2 "$T/host.log"
N2=$(grep -c "runtime error\|ERROR: AddressSanitizer" "$T/host.log")
echo "sanitizer reports: oracle $N1, library host sources $N2"
[ "$N1" = 0 ] && [ "$N2" = 0 ]
