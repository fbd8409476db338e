for cfg in "12 512" "12 256" "12 128" "13 512" "13 256" "14 512"; do
  set -- $cfg
  export FQD_SEG_BITS=$1 FQD_DEDUP_THREADS=$2
  python bench.py --steps 3 --warmup 1 --cpu-sample 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg', d['value'], d['ms_per_step'], d['roofline']['kernels'])"
done
