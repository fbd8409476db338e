import sys, time
sys.path.insert(0, ".")
import torch
from fastq_dupaway_amd import Engine, Reads
n, L = 50_000_000, 150
dev = torch.device("cuda", 0)
bases = torch.empty(n * L + 16, dtype=torch.uint8, device=dev)
keep = torch.empty(n, dtype=torch.uint8, device=dev)
for no_stage in (False, True):
    with Engine(segments=1, capacity_reads=n, capacity_bases=n * L, profile=True, no_stage=no_stage) as e:
        e.synth_reads(1, 0, n, L, 200, 0, bases, None); e.sync()
        for _ in range(3):
            e.reset(); e.submit([Reads(bases, uniform_len=L, uniform_stride=L)], n, keep=keep); e.sync()
        p = e.profile()
        print("no_stage", no_stage, "encode avg ms", p["encode_ms"] / p["encode_launches"], "GB/s algorithmic", n * L / (p["encode_ms"] / p["encode_launches"] * 1e-3) / 1e9)
# ragged descriptor over the same data (offsets/lengths arrays) -> general kernel with explicit offsets
offs = (torch.arange(n, dtype=torch.int64, device=dev) * L)
lens = torch.full((n,), L, dtype=torch.int32, device=dev)
with Engine(segments=1, capacity_reads=n, capacity_bases=n * L, profile=True) as e:
    for _ in range(3):
        e.reset(); e.submit([Reads(bases, offs, lens)], n, keep=keep); e.sync()
    p = e.profile()
    print("ragged desc: encode avg ms", p["encode_ms"] / p["encode_launches"], "insert/partition/dedup", p["insert_ms"], p["partition_ms"] / max(1, p["partition_launches"]), p["dedup_ms"] / max(1, p["dedup_launches"]))
