cd sim2real_lane_segment_amd/csrc && rm -f build/dense3.o build/net.o && bash build.sh -DRLN_DIAG > /dev/null 2>&1; cd ../..
for d in 0 1 2 4 3 5 6 7; do RLN_D3_DBG=$d python tools/d3_fwd_bench.py 2 1 2>&1 | grep -E "cin  112|cin  272"; done
