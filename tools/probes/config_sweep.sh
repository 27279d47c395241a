# does every neighbouring configuration of the benchmark still run? (precision, batch, node count, ragged batches, 16-wide kernels)
for a in "--precision fp16" "--precision fp32 --batch 64" "--nodes 16" "--nodes 24 --batch 100 --ragged" "--batch 8" "--nodes 48 --batch 64 --ragged" "--nodes 64 --batch 32" "--nodes 40 --batch 96 --precision fp16"; do
  python bench.py --no-cpu-baseline --steps 5 --warmup 2 --settle-steps 6 $a 2>/tmp/err.txt | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('[$a]', d['value'], 'graphs/s', d['ms_per_step'], 'ms  loss', d['final_loss'], ' roofline kernel', d['roofline'].get('kernel'), d['roofline'].get('frac'))"
  tail -2 /tmp/err.txt | grep -i -E "error|Traceback" | head -2
done
