run() { FQD_BENCH_FORCE_SHARDED=1 timeout -k 10 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 3 --warmup 1 --cpu-sample 0 "$@" 2>&1 | grep -E "^\{|differ|Error" | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l); print(d['value'], d['ms_per_step'], d['config']['sharding'], d['parity'][:30])
    except Exception: print(l[:200])"; }
echo pipe=1; FQD_SHARDED_PIPELINE=1 run
echo pipe=0; FQD_SHARDED_PIPELINE=0 run
echo pipe=1 paired 30M; FQD_SHARDED_PIPELINE=1 run --paired --reads 30000000
echo lazy; FQD_SHARDED_LAZY=1 run
echo lazy rounds=2; FQD_SHARDED_LAZY=1 FQD_BENCH_ROUNDS=2 run
echo lazy paired 30M; FQD_SHARDED_LAZY=1 run --paired --reads 30000000
