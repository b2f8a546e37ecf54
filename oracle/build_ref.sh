#!/bin/bash
# Builds the reference minicom (read-only sources under /root/reference/src) into oracle/_ref/.
# TEST INFRASTRUCTURE ONLY.  Nothing is copied from the reference: sources are compiled where they lie.
# The reference's own CLI writes src/config.h on every run (reference minicom:56-91); this recipe
# writes the same macro set into oracle/_ref/cfg_<variant>/config.h and compiles with g++ directly
# (the reference Makefile is not run).  Outputs only under oracle/_ref/ (git-ignored, travels via gpurun).
set -e
REF=${MCOM_REFERENCE:-/root/reference}/src
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=$HERE/_ref
[ -d "$REF" ] || { echo "reference sources not present at $REF: skipping oracle/_ref build"; exit 0; }
mkdir -p "$OUT"
CXX=${CXX:-g++}
# -march: the golden-fixture variants are built for x86-64-v2 (any box can run them; their outputs do not depend on it); the
# CPU-baseline variants (_t*) for x86-64-v3 -- the reference's own Makefile:2 says -march=native, which cannot travel to another
# box; v3 (AVX2, BMI2, POPCNT) is what every current server CPU offers and is what bench.py names in cpu_baseline.sample
base_flags() { echo "-O3 -Wno-unused-function -std=c++11 -w -march=${MARCH:-x86-64-v2} -fopenmp"; }
LIB_SRCS="bseq misc preprocess sketch bbhashdict kthread_reads kthread_bucket kthread_idx kthread_cb kthread_hash_realign kthread_dump "

build_variant() {   # name readlen extra_defines...
  local name=$1 L=$2; shift 2
  local d=$OUT/$name
  local CXXFLAGS; CXXFLAGS=$(base_flags)
  mkdir -p "$d/output_ref"
  if [ -f "$d/.march" ] && [ "$(cat "$d/.march")" != "${MARCH:-x86-64-v2}" ]; then rm -f "$d"/*.o; fi
  echo "${MARCH:-x86-64-v2}" > "$d/.march"
  {
    echo "#pragma once"
    for x in "$@"; do
      echo "#define $x"
      case "$x" in ORDER|_PE) echo "int cmpcluster3(const void *a_, const void *b_);";; esac
    done
    echo "#define readlen $L"
    echo "#define num_thr ${NUM_THR:-1}"
    echo "#define uniqid \"uref\""
    echo "#define output \"output_ref/\""
    for m in inik inithr inimaxthr inistep iniw inim inicbthr inimaxrounds; do echo "#define $m 0"; done
    echo "#define ininumdict ${ININUMDICT:-0}"       # -s is compiled in (kthread_hash_realign.c:153); the others reach refdump as arguments
  } > "$d/config.h"
  local defs=""
  for x in "$@"; do defs="$defs -D$x"; done
  local srcs="$LIB_SRCS"
  # the reference links through an archive, so only one of the two dump objects is ever pulled in
  case " $* " in *" _PE "*) srcs="${srcs/kthread_dump /} kthread_dump_pe";; esac
  local objs=""
  for s in $srcs; do
    if [ ! -f "$d/$s.o" ] || [ "$REF/$s.c" -nt "$d/$s.o" ]; then
      $CXX -c $CXXFLAGS $defs -I"$d" -I"$REF" "$REF/$s.c" -o "$d/$s.o" &
    fi
    objs="$objs $d/$s.o"
  done
  wait
  # the reference's main(), renamed so the same object supplies the globals to the dumper
  $CXX -c $CXXFLAGS $defs -I"$d" -I"$REF" "$REF/minicommain.c" -o "$d/minicommain.o"
  $CXX -c $CXXFLAGS $defs -Dmain=ref_main -I"$d" -I"$REF" "$REF/minicommain.c" -o "$d/minicommain_nomain.o"
  $CXX $CXXFLAGS "$d/minicommain.o" $objs -o "$d/minicom_bin" -lm -lz -lpthread
  $CXX -c $CXXFLAGS $defs -I"$d" -I"$REF" "$REF/decompress.c" -o "$d/decompress.o"
  $CXX $CXXFLAGS "$d/decompress.o" -o "$d/decompress" -lm -lz -lpthread
  # golden-vector dumper (our code, oracle/refdump.cpp) linked against the reference objects
  $CXX $CXXFLAGS $defs -I"$d" -I"$REF" "$HERE/refdump.cpp" "$d/minicommain_nomain.o" $objs -o "$d/refdump" -lm -lz -lpthread
  echo "built $d"
}

# INTEGRATION.md section A as a link: the reference's own main / pre_process / cluster_dump / reader objects of a variant built above,
# with oracle/shim_stages.cpp (the four stage drivers over libmcom_host.so) in the place of the reference's kthread_reads / _bucket /
# _cb / _hash_realign / bbhashdict objects -> oracle/_ref/<variant>/minicom_gpu (needs the product libraries: skipped when they are not built)
build_shim() {   # variant
  local d=$OUT/$1
  local LIBD=$HERE/../minicom_amd/lib
  [ -f "$LIBD/libmcom_host.so" ] && [ -f "$LIBD/libmcom_hip.so" ] || { echo "libmcom_host.so not built: no $1/minicom_gpu"; return 0; }
  local CXXFLAGS; CXXFLAGS=$(base_flags)
  $CXX -c $CXXFLAGS -I"$d" -I"$REF" -I"$HERE" "$HERE/shim_stages.cpp" -o "$d/shim_stages.o"
  $CXX $CXXFLAGS "$d/minicommain.o" "$d/preprocess.o" "$d/kthread_dump.o" "$d/bseq.o" "$d/misc.o" "$d/sketch.o" "$d/kthread_idx.o" "$d/shim_stages.o" \
       -o "$d/minicom_gpu" -L"$LIBD" -lmcom_host -lmcom_hip -Wl,-rpath,'$ORIGIN/../../../minicom_amd/lib' -lm -lz -lpthread
  echo "built $d/minicom_gpu"
}

build_variant L100 100
build_variant L150 150
build_shim L100
build_shim L150
build_variant L40 40
build_variant L75 75
ININUMDICT=4 build_variant L100_s4 100
build_variant L100_order 100 ORDER
build_variant L100_pe 100 _PE
build_variant L150_order 150 ORDER
build_variant L150_pe 150 _PE
# multi-threaded builds, used only as the CPU baseline of bench.py (their output is not reproducible run to run)
MARCH=x86-64-v3 NUM_THR=16 build_variant L150_t16 150
# ... at other core counts: bench.py takes the largest one the box has cores for (north_star: "-t <host cores>")
MARCH=x86-64-v3 NUM_THR=8 build_variant L150_t8 150
MARCH=x86-64-v3 NUM_THR=32 build_variant L150_t32 150
MARCH=x86-64-v3 NUM_THR=64 build_variant L150_t64 150
# the same at x86-64-v2, should a box lack AVX2 (bench.py falls back to it and says so)
NUM_THR=64 build_variant L150_t64_v2 150
