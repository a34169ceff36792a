run() { timeout -k 10 200 python bench.py $X --steps 20 --warmup 3 --no-cpu-baseline --no-latent --no-elided --no-mnist --no-roofline --sample-images 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['value']), round(d['ms_per_step'],3))"; }
X="" run plain_q4
X="--rehearse-dp" DM_RESERVE_MODE=nomain run q4_reserve_nomain
X="--rehearse-dp" run q4_reserve_mainfirst
X="--rehearse-dp" GPU_MAX_HW_QUEUES=8 DM_RESERVE_MODE=nomain run q8_reserve_nomain
X="--rehearse-dp" GPU_MAX_HW_QUEUES=8 run q8_reserve_mainfirst
X="--rehearse-dp" GPU_MAX_HW_QUEUES=8 DM_RESERVE_STREAMS=0 run q8_noreserve
X="--rehearse-dp" GPU_MAX_HW_QUEUES=6 run q6_reserve_mainfirst
X="--rehearse-dp" GPU_MAX_HW_QUEUES=3 run q3_reserve_mainfirst
