#!/bin/bash
# Builds a variant of the library into build_ab/<name>/liborbx.so with extra compiler flags, for
# tools/ab_lib.sh; the in-tree objects and library are rebuilt without the flags afterwards.
# usage: tools/build_variant.sh <name> <source.hip> [flags...]
set -e
cd "$(dirname "$0")/.."
NAME=$1; SRC=$2; shift 2
mkdir -p build_ab/$NAME
touch orbslam2_amd/csrc/$SRC
make -s -C orbslam2_amd/csrc CXXFLAGS="-O3 -fPIC -std=c++17 -ffp-contract=off -Wall -Wno-unused-result -I../../include -I. $*"
cp orbslam2_amd/liborbx.so build_ab/$NAME/
touch orbslam2_amd/csrc/$SRC
make -s -C orbslam2_amd/csrc
