#!/bin/bash
# builds libtpg_hip.so variants that differ by compile-time macros into abvar/ (A/B timing inside one GPU job with
# tools/lib_ab.py / tools/enc_ab.py; abvar/ is listed in .gpurunignore so that stale variants do not ship: take the line out
# for the job that needs them):  tools/build_variants.sh name1 "-DX=1" name2 "-DX=2" ...
set -e
cd "$(dirname "$0")/.."
mkdir -p abvar
while [ $# -gt 1 ]; do
  name=$1; flags=$2; shift 2
  tmp=$PWD/tidypopgen_amd/.ab_$name   # a sibling of csrc/: the sources' relative includes resolve
  rm -rf "$tmp"; mkdir -p "$tmp"
  cp -r tidypopgen_amd/csrc/*.hip tidypopgen_amd/csrc/*.h tidypopgen_amd/csrc/host tidypopgen_amd/csrc/Makefile "$tmp"/
  sed -i "s#^CXXFLAGS = #CXXFLAGS = $flags #; s#^OUT = .*#OUT = $PWD/abvar/libtpg_$name.so#" "$tmp/Makefile"
  make -C "$tmp" -j8 -s
  rm -rf "$tmp"
  echo "built abvar/libtpg_$name.so ($flags)"
done
