tools/abw.sh "--workload histogram" ab_libs/lib_seg1.so ab_libs/lib_seg2.so ab_libs/lib_seg4.so ab_libs/lib_seg8.so ab_libs/lib_seg3.so ab_libs/lib_seg12.so whisper_amd/libwhisper_gfx950.so
echo prefetch; WH_CHAIN_PREFETCH=1 tools/abw.sh "--workload histogram" whisper_amd/libwhisper_gfx950.so
WH_CHAIN_PREFETCH=1 tools/abw.sh "--workload average" whisper_amd/libwhisper_gfx950.so
WH_CHAIN_PREFETCH=1 tools/abw.sh "--workload sumvec_cfg1" whisper_amd/libwhisper_gfx950.so
