tools/abw.sh "--workload histogram" whisper_amd/libwhisper_gfx950.so ab_libs/lib_qt4.so
