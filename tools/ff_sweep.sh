for q in "" 8; do for l in 1 2; do for f in 1 2 3 4; do
  r=$(env MCRT_LANES=$l ${q:+GPU_MAX_HW_QUEUES=$q} timeout -k 10 200 python bench.py --no-cpu-baseline --frames-in-flight $f --steps 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['latency_ms'])")
  echo "hwq=${q:-default} lanes=$l F=$f: $r"
done; done; done
