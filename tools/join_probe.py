#!/usr/bin/env python3
"""Times the device tag join (fqd_join_tags) on synthetic tags and checks it in closed form.
  python tools/join_probe.py <n_ids> <one_word|two_words|illumina> [reps]
Prints a line per stage as it goes (append-friendly for gpurun_out logs)."""
import sys, time
sys.path.insert(0, ".")
import torch
from fastq_dupaway_amd import Engine

NONE = 0xFFFFFFFF


def log(*a):
    print(f"[{time.strftime('%H:%M:%S')}]", *a, flush=True)


def digits(x, width):
    cols = []
    for _ in range(width):
        cols.append((x % 10 + 48).to(torch.uint8)); x = x // 10
    return torch.stack(cols[::-1], dim=1)


def main():
    n_ids = int(sys.argv[1]); style = sys.argv[2]; reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    dev = torch.device("cuda")
    g = torch.Generator(device=dev); g.manual_seed(77)
    ids = torch.arange(n_ids, dtype=torch.int64, device=dev)
    if style == "one_word":
        key = ids
        body = torch.cat([torch.full((n_ids, 1), ord("r"), dtype=torch.uint8, device=dev), digits(ids, 9),
                          torch.full((n_ids, 1), 10, dtype=torch.uint8, device=dev)], dim=1)
    else:
        f1 = (ids * 2654435761 + 12345) % 1_000_000_000     # 9 digits: f1 * 1e9 + ids stays inside int64
        key = f1 * 1_000_000_000 + ids
        prefix = torch.tensor(list(b"M01234:55:000000000-ABCDE:1:"), dtype=torch.uint8, device=dev).repeat(n_ids, 1)
        body = torch.cat([prefix, digits(f1, 9), torch.full((n_ids, 1), ord("#"), dtype=torch.uint8, device=dev), digits(ids, 9)], dim=1)
        del prefix
    width = body.shape[1]
    log("tags built", style, "width", width)

    def side(rem):
        mine = ids[ids % 10 != rem]
        return mine[torch.randperm(mine.numel(), device=dev, generator=g)]
    ida, idb = side(3), side(7)
    na, nb = ida.numel(), idb.numel()
    tags_a = body[ida].contiguous().view(-1); tags_b = body[idb].contiguous().view(-1)
    del body
    off_a = torch.arange(na, dtype=torch.int64, device=dev) * width; off_b = torch.arange(nb, dtype=torch.int64, device=dev) * width
    len_a = torch.full((na,), width, dtype=torch.int32, device=dev); len_b = torch.full((nb,), width, dtype=torch.int32, device=dev)
    i32 = dict(dtype=torch.int32, device=dev)
    pa, ma, pb, mb = torch.empty(na, **i32), torch.empty(na, **i32), torch.empty(nb, **i32), torch.empty(nb, **i32)
    qa, qb = torch.empty(min(na, nb), **i32), torch.empty(min(na, nb), **i32)
    torch.cuda.synchronize()
    log("sides built", na, nb)
    with Engine(segments=2) as e:
        for r in range(reps):
            t0 = time.perf_counter()
            n_pairs = e.join_tags((tags_a, off_a, len_a, na), (tags_b, off_b, len_b, nb), pa, pb, ma, mb, qa, qb)
            dt = time.perf_counter() - t0
            log(f"join rep {r}: {dt * 1e3:.1f} ms  ({(na + nb) / dt / 1e6:.0f} M tags/s)  pairs {n_pairs}")
    exp_pa = torch.argsort(key[ida]); exp_pb = torch.argsort(key[idb])
    log("perm_a ok", bool(torch.equal(pa.long(), exp_pa)), "perm_b ok", bool(torch.equal(pb.long(), exp_pb)))
    common = (ids % 10 != 3) & (ids % 10 != 7)
    log("n_pairs ok", n_pairs == int(common.sum()))
    sa, sb = ida[exp_pa], idb[exp_pb]
    has_a = common[sa]; has_b = common[sb]
    log("match_a presence ok", bool(torch.equal(ma.long() != -1, has_a)))
    log("pairs ok", bool(torch.equal(ida[qa[:n_pairs].long()], sa[has_a])), bool(torch.equal(idb[qb[:n_pairs].long()], sb[has_b])))
    k = torch.nonzero(has_a)[:, 0]
    log("partners ok", bool(torch.equal(mb[ma[k].long()].long(), k)))


if __name__ == "__main__":
    main()
