import sys, time
sys.path.insert(0, ".")
import torch
from fastq_dupaway_amd import Engine, Reads
n, L = 100_000_000, 150
dev = torch.device("cuda", 0)
bases = torch.empty(n * L + 16, dtype=torch.uint8, device=dev)
expect = torch.empty(n, dtype=torch.uint8, device=dev); keep = torch.empty(n, dtype=torch.uint8, device=dev)
eng = Engine(segments=1, device=0, capacity_reads=n, capacity_bases=n * L, profile=True)
eng.synth_reads(2026, 0, n, L, 200, 0, bases, expect); eng.sync()
for parts in (1, 2, 4, 1, 2):
    m = n // parts
    def step():
        eng.reset()
        for k in range(parts):
            eng.submit([Reads(bases[k * m * L:], uniform_len=L, uniform_stride=L)], m, keep=keep[k * m:])
        eng.sync()
    step(); step()
    assert bool(torch.equal(keep, expect))
    eng.reset_profile(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(6): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 6 * 1e3
    p = eng.profile()
    print(f"parts={parts}: {dt:.3f} ms/step  encode {p['encode_ms']/6:.3f} partition {p['partition_ms']/6:.3f} dedup {p['dedup_ms']/6:.3f} insert {p['insert_ms']/6:.3f} other {p['other_ms']/6:.3f}", flush=True)
