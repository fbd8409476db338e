for rep in 1 2; do for f in 1 0; do
  FQD_FOLD_HIST1=$f python bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-verify 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fold=$f', d['value'], d['ms_per_step'], {k:v['avg_ms'] for k,v in d['roofline']['kernels'].items()})"
done; done
