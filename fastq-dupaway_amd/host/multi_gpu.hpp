// multi_gpu.hpp — one dedup job over several MI355X from the C++ host driver.
// The reference has nothing to mirror here (it is single-threaded, single-process); this is the
// north_star's "reads partitioned across the GPUs of one node by hash prefix with an RCCL all-to-all
// over xGMI so each GPU owns a disjoint bucket range", driven through the same C ABI halves
// (include/fqdupaway.h: fqd_encode_uniform / fqd_partition_keys / fqd_reserve_keys / fqd_insert_keys /
// fqd_scatter_flags) that fastq-dupaway_amd/sharded.py drives from Python for bench.py.
//
// One process, one engine per GPU.  A round deals the next batches of the input to the ranks in
// order (global input order = round, rank, position = file order); every rank encodes its batch
// and groups the keys by owner = hash prefix mod ranks; ONE all-to-all moves the keys to their
// owners, which insert them in (source rank, position) order — so first-occurrence-wins is global —
// and the keep flags travel back through the reverse all-to-all.
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>
#include <vector>

#include <hip/hip_runtime_api.h>

namespace fqdhost {

// Offsets of one round's all-to-all, in records.  send(s,d) comes from the ranks' partition counts.
struct ExchangePlan {
    int n = 0;
    std::vector<uint64_t> send;       // [s*n + d] records rank s sends to rank d
    std::vector<uint64_t> send_off;   // [s*n + d] where they start in s's grouped buffer (grouped by d, ascending)
    std::vector<uint64_t> recv_off;   // [d*n + s] where they land in d's receive buffer (sources in rank order)
    std::vector<uint64_t> n_send;     // [s] records s sends in all
    std::vector<uint64_t> n_recv;     // [d] records d receives in all
    explicit ExchangePlan(int ranks) : n(ranks), send(size_t(ranks) * ranks, 0) {}
    void finish();                    // fills the offsets and totals from send[]
};

// One byte range to move from a buffer on one rank's GPU to a buffer on another's.
struct Transfer { int src_rank; const void* src; int dst_rank; void* dst; size_t bytes; };

// Builds the transfers of an exchange: forward = sources' grouped buffers -> owners' receive
// buffers; backward = owners' per-record results (receive layout) -> sources (grouped layout).
std::vector<Transfer> forward_transfers(const ExchangePlan& p, const std::vector<const void*>& grouped,
                                        const std::vector<void*>& received, size_t item_bytes);
std::vector<Transfer> backward_transfers(const ExchangePlan& p, const std::vector<const void*>& at_owner,
                                         const std::vector<void*>& at_source, size_t item_bytes);

// Executes transfers between the ranks' GPUs.  The caller makes sure the sources are complete
// before run() and waits for every rank's stream after it.
class Exchange {
public:
    virtual ~Exchange() = default;
    virtual void run(const std::vector<Transfer>& t) = 0;
    virtual const char* name() const = 0;
    // RCCL over xGMI (ncclSend/ncclRecv grouped into one all-to-all) when every rank has its own
    // GPU and `prefer_rccl`; otherwise peer copies (hipMemcpyPeerAsync; plain device copies between
    // ranks that share a GPU — rehearsals of N ranks on one card).
    static std::unique_ptr<Exchange> create(const std::vector<int>& devices, const std::vector<hipStream_t>& streams, bool prefer_rccl);
};

// FQD_DEVICES="0,1,2,3" -> {0,1,2,3}; empty when unset.  A device may repeat (virtual ranks).
std::vector<int> devices_from_env();

} // namespace fqdhost
