# same-box comparison: hardware queues per process (GPU_MAX_HW_QUEUES, HIP default 4) for the plain step and the one-rank RCCL rehearsal
run() { timeout -k 10 200 python bench.py $X --steps 20 --warmup 3 --no-cpu-baseline --no-latent --no-elided --no-mnist --no-roofline --sample-images 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['value']), round(d['ms_per_step'],3))"; }
for i in 1 2; do
X="" run plain_q4
X="" GPU_MAX_HW_QUEUES=8 run plain_q8
X="--rehearse-dp" run dp_q4
X="--rehearse-dp" GPU_MAX_HW_QUEUES=8 run dp_q8
X="--rehearse-dp" DM_RESERVE_STREAMS=0 GPU_MAX_HW_QUEUES=8 run dp_q8_noreserve
done
