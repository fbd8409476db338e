set -e
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "full_size or synthetic" 2>&1 | tail -3
for cfg in "4194304 4 4" "4194304 4 8" "4194304 8 8" "8388608 4 4" "2097152 4 4" "4194304 3 5" "0 8 8"; do
  set -- $cfg
  if [ "$1" = "0" ]; then export FQD_CHUNK_READS=99999999999; else export FQD_CHUNK_READS=$1; fi
  export FQD_ENC_BLOCKS_PER_CU=$2 FQD_INS_BLOCKS_PER_CU=$3
  python bench.py --steps 3 --warmup 1 --cpu-sample 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg', d['value'], d['ms_per_step'], d['roofline']['kernels'])"
done
