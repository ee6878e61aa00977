#!/bin/bash
# Build a variant of the library for an A/B run: tools/variant.sh <name> [compiler flags ...]  ->  ab_libs/lib_<name>.so
# (objects under build/var_<name>; select it at run time with WHISPER_GFX950_LIB, as tools/abw.sh does)
name=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/ab_libs
make -s -C $R/whisper_amd/csrc -j8 OUT=$R/ab_libs/lib_$name.so OBJDIR=$R/build/var_$name EXTRA="$*" 2>&1 | grep -E "error|Error" 
ls -la $R/ab_libs/lib_$name.so | awk '{print $5, $9}'
