# sampling throughput against the number of images per batch (UNet batch = 2x: cond + uncond); 100 of the 500 steps
for n in 16 32 44 64 88 128 176; do
timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-latent --no-elided --no-mnist --no-roofline --sample-images $n --sample-steps 100 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['sampling']; print('images', s['images'], 'img/s (500-step equivalent)', round(s['value']*s['steps']/500,2))"
done
