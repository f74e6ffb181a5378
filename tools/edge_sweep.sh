for ws in 1024 4096 16384; do for n in 64 256 1024; do echo "ws=$ws n=$n smooth"; PSD_EDGE_WS_MB=$ws ET_N=$n ET_SMOOTH=1 timeout 120 python tools/edge_time.py 2>&1 | tail -1; done; done
echo "uniform"; PSD_EDGE_WS_MB=4096 ET_N=256 timeout 120 python tools/edge_time.py 2>&1 | tail -1
