#!/bin/bash
# tools/ktrace_lib.sh lib.so N : N kernel timelines of one bench step with the given library
L=$1; N=${2:-3}
for i in $(seq $N); do WHISPER_GFX950_LIB=$L tools/ktrace.sh | grep -v "k_agg\|k_keep\|redo" || exit 1; echo; done
