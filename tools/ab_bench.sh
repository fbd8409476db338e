#!/bin/bash
# A/B of two builds of the engine library on one GPU box, runs interleaved so that box-to-box and
# run-to-run drift cancels:  tools/ab_bench.sh <lib_a.so> <lib_b.so> [rounds] [bench.py args...]
# (build the variants with `make -C fastq-dupaway_amd lib` and copy lib/libfqdupaway.so aside).
A=$1; B=$2; N=${3:-3}; shift 3 2>/dev/null
LIB=fastq-dupaway_amd/lib/libfqdupaway.so
cp $LIB /tmp/ab_keep.so
for i in $(seq $N); do for v in A B; do
  if [ $v = A ]; then cp $A $LIB; else cp $B $LIB; fi
  echo -n "$v: "; timeout -k 10 300 python bench.py --steps 6 --warmup 2 --cpu-sample 0 "$@" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); k=d['roofline']['kernels']; print(d['ms_per_step'], {a:k[a]['avg_ms'] for a in k})"
done; done
cp /tmp/ab_keep.so $LIB
