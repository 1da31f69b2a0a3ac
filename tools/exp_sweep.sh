for W in 2 4; do echo "wpb=$W"; M2S_WPB=$W python bench.py --steps 6 --warmup 2 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['phases_ms']['distance_per_launch'])"; done
for mb in 16 32 64 128; do echo "piece_mb=$mb"; M2S_HOST_PIECE_MB=$mb python tools/exp_host_calls.py 2>&1 | grep -E "512\^3|256\^3" | cut -c1-120; done
