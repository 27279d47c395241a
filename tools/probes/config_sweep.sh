for a in "--precision fp16" "--precision fp32 --batch 64" "--nodes 16" "--nodes 24 --batch 100 --ragged" "--batch 8"; do
  python bench.py --no-cpu-baseline --steps 5 --warmup 2 $a 2>/tmp/err.txt | tail -1 | cut -c1-230 | sed "s|^|[$a] |"
  tail -2 /tmp/err.txt | grep -i -E "error|Traceback" | head -2
done
