# same-box comparison of the training leg: plain, one-rank RCCL rehearsal and its variants (developer tool)
run() { timeout -k 10 200 python bench.py $X --steps 20 --warmup 3 --no-cpu-baseline --no-latent --no-elided --no-mnist --no-roofline --sample-images 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['value']), round(d['ms_per_step'],3))"; }
for i in 1 2; do
X="" run plain
X="--rehearse-dp" run dp_async
X="--rehearse-dp" DM_DP_SYNC=1 run dp_sync
X="--rehearse-dp" DM_REHEARSE_MODE=pgonly run pgonly
done
