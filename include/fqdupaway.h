/* fqdupaway.h — C ABI of the MI355X-native `--fast` deduplication engine.
 *
 * The reference (fastq-dupaway V1.5.0) has no FFI: its seam for this path is the
 * C++ class template HashDupRemover<T> (src/hash_dup_remover.hpp:73-94), which
 * builds one key per record (setRecord / setRecordPair, hpp:19-41 via
 * SeqUtils::seq2hash, src/seq_utils.cpp:35-49) and runs find-then-insert on a
 * std::unordered_set (hpp:70-71,126-144,228-248).  This header is what a
 * binding for that loop would call instead: the host side keeps parsing
 * records and writing survivors (fastq-dupaway_amd/host mirrors
 * HashDupRemover's constructor, filterSE and filterPE), and hands batches of
 * sequences to the device, which answers with one keep flag per record.
 *
 * Conventions: plain pointers and sizes only; every function returns an int
 * status (FQD_OK = 0) and never throws; the text of the last failure of an
 * engine is available from fqd_last_error().  `device` pointers are HIP
 * device pointers, `stream` is a hipStream_t passed as void*.
 *
 * Contract: record i of an engine's input
 * order is KEPT iff no record j < i has the identical sequence (for paired
 * engines: identical mate-1 AND mate-2 sequences, lengths included) over the
 * alphabet {A,C,G,T,N}.  Selection is exact — hashes only place keys — and
 * the first occurrence always wins.  Any other byte is an error that carries
 * the offending byte, as SeqUtils::_char2number does (seq_utils.cpp:3-21).
 */
#ifndef FQDUPAWAY_H
#define FQDUPAWAY_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FQD_ABI_VERSION 5

/* status codes */
#define FQD_OK              0
#define FQD_ERR_ARG         1   /* bad argument / misuse                              */
#define FQD_ERR_HIP         2   /* a HIP runtime call failed (see fqd_last_error)     */
#define FQD_ERR_BAD_BASE    3   /* a byte outside {A,C,G,T,N}: see fqd_bad_base()     */
#define FQD_ERR_CAPACITY    4   /* more than 2^32-2 records in one engine             */
#define FQD_ERR_NO_DEVICE   5   /* no usable MI355X / HIP device                      */

/* where the caller's buffers live */
#define FQD_MEM_HOST        0
#define FQD_MEM_DEVICE      1

/* fqd_config.flags */
#define FQD_FLAG_PROFILE    1u  /* bracket every kernel with HIP events (fqd_profile) */
#define FQD_FLAG_NO_STAGE   2u  /* testing: force the per-lane global-load encoder    */
#define FQD_FLAG_WEAK_HASH  4u  /* testing: zero every hash's tag and its low 6 position bits,
                                   so unequal keys keep meeting in the table and the
                                   verify-then-probe-on branch runs on every insert    */

typedef struct fqd_engine fqd_engine;

/* Replaces: HashDupRemover<T>::HashDupRemover(memlimit, tempdir, verbose)
 * (hash_dup_remover.hpp:77-78) + the `hashed_set records; records.reserve(ONE_MIL)`
 * of each driver (hpp:113-114,206-207,269-270). */
typedef struct fqd_config {
    int32_t  device;          /* HIP device ordinal                                       */
    int32_t  segments;        /* 1: setRecord keys (SE); 2: setRecordPair keys (PE)       */
    uint64_t capacity_reads;  /* hint: records expected over the engine's life (0 = grow) */
    uint64_t capacity_bases;  /* hint: total bases expected (0 = grow)                    */
    void*    stream;          /* hipStream_t to run on; NULL = the engine creates one     */
    uint32_t flags;           /* FQD_FLAG_*                                               */
    uint32_t reserved;
} fqd_config;

/* One mate's sequences for a batch: ASCII, one byte per base, no newlines
 * needed between reads.  Either ragged (offsets + lengths arrays, in the same
 * memory space as `bases`) or uniform (offsets == lengths == NULL: read i is
 * the uniform_len bytes at bases + i * uniform_stride).
 * Replaces the (obj.seq(), obj.seq_len()-1) pairs handed to the key
 * constructors at hash_dup_remover.hpp:124,131,223-224,234-235,293-294. */
typedef struct fqd_reads {
    const uint8_t*  bases;
    const uint64_t* offsets;
    const uint32_t* lengths;
    uint32_t        uniform_len;
    uint32_t        uniform_stride;
} fqd_reads;

typedef struct fqd_stats {
    uint64_t records;         /* records submitted so far (tot_reads, hpp:116)         */
    uint64_t duplicates;      /* keep flags cleared so far (dup_reads, hpp:116)        */
    uint64_t table_slots;     /* current capacity of the hash set                      */
    uint64_t key_bytes;       /* bytes of packed keys resident in HBM                  */
} fqd_stats;

/* Average device time per launch, measured with HIP events on the engine's
 * stream when FQD_FLAG_PROFILE is set (bench.py's roofline uses these). */
typedef struct fqd_profile {
    double   encode_ms;    uint64_t encode_launches;    uint64_t encode_reads;     /* encode_*_kernel            */
    double   insert_ms;    uint64_t insert_launches;    uint64_t insert_reads;     /* insert_kernel (atomic path) */
    double   partition_ms; uint64_t partition_launches; uint64_t partition_reads;  /* bulk path: hist + scatter passes, timed as one group per batch */
    double   dedup_ms;     uint64_t dedup_launches;     uint64_t dedup_reads;      /* bulk path: bucket_dedup_kernel */
    double   other_ms;     uint64_t other_launches;                                /* clears, scans, rehash      */
} fqd_profile;

int  fqd_abi_version(void);
int  fqd_device_count(int* count);

int  fqd_engine_create(const fqd_config* cfg, fqd_engine** out);
int  fqd_engine_destroy(fqd_engine* e);
/* Empties the set and the key store, keeping the allocations. */
int  fqd_engine_reset(fqd_engine* e);

/* Replaces one pass of the hot loop (hpp:126-144 SE, 228-248 PE) over `n`
 * records: builds their keys, probes/inserts them, and writes keep[i] = 1 if
 * record i (input order continues across calls) is the first with its key,
 * else 0.  `seg` points at cfg.segments descriptors.  `memory` says where
 * bases/offsets/lengths/keep live.  Device-space calls are asynchronous on
 * the engine's stream (the buffers must be complete on it: see the ORDERING RULE
 * at fqd_engine_wait_stream) and keep must stay valid until fqd_engine_sync();
 * host-space calls return with keep filled.  Flags of a batch are final when
 * it completes: later batches cannot change them. */
int  fqd_submit(fqd_engine* e, const fqd_reads* seg, uint64_t n, int memory, uint8_t* keep);

/* fqd_submit for the LAST batch of a run: the caller declares that nothing will be added before the next
 * fqd_engine_reset.  The reference's loop produces flags, not a set anyone looks at afterwards
 * (hash_dup_remover.hpp:126-144): the bulk insert path then leaves the set's segments where they were built
 * (on chip) instead of writing them back to HBM — 8 bytes per slot, 2 GiB for the 100 M-read table.  Same flags as
 * fqd_submit, bit for bit.  Afterwards fqd_submit* / fqd_insert_* fail with FQD_ERR_ARG until the engine is reset;
 * stats, flags and fqd_bad_base work as after fqd_submit. */
int  fqd_submit_final(fqd_engine* e, const fqd_reads* seg, uint64_t n, int memory, uint8_t* keep);

/* The hipStream_t the engine launches on (its own, or the one given in fqd_config), so a caller
 * can order its own work — copies, collectives — against the engine's with events. */
void* fqd_engine_stream(fqd_engine* e);

/* ORDERING RULE.  The engine runs on its own non-blocking stream (unless fqd_config.stream gave it the caller's) and
 * waits for nobody: a device buffer handed to any entry point of this header must be COMPLETE on the engine's stream
 * when the call is made — a fill, copy or kernel the caller queued on another stream (the null stream included) is
 * not ordered against the engine's work by itself and may land after it.  Two ways to order, besides a device-wide
 * synchronise: fqd_engine_wait_stream() before handing buffers over ("the engine starts after everything queued on
 * `stream` so far"), and fqd_stream_wait_engine() before the caller's stream reads what the engine wrote ("`stream`
 * continues after everything the engine has queued so far").  Both are an event record plus a stream wait: nothing
 * blocks on the host.  `stream` is a hipStream_t; NULL = the null stream. */
int  fqd_engine_wait_stream(fqd_engine* e, void* stream);
int  fqd_stream_wait_engine(fqd_engine* e, void* stream);

/* Waits for the stream and surfaces deferred errors (FQD_ERR_BAD_BASE). */
int  fqd_engine_sync(fqd_engine* e);

/* After FQD_ERR_BAD_BASE: the first offending byte in input order — record
 * index, segment (0/1), position in that sequence, and the byte itself — so a
 * caller can print the reference's two lines (seq_utils.cpp:18-19).  Keep flags
 * of records before `record` are valid. */
int  fqd_bad_base(const fqd_engine* e, uint64_t* record, uint32_t* segment, uint32_t* position, uint8_t* byte);

int  fqd_get_stats(fqd_engine* e, fqd_stats* out);
int  fqd_get_profile(fqd_engine* e, fqd_profile* out);
int  fqd_reset_profile(fqd_engine* e);
const char* fqd_last_error(const fqd_engine* e);   /* NULL engine: last create() failure */

/* ---- the two halves of fqd_submit, exposed for multi-GPU sharding ----------
 * (encode where the reads are, exchange fixed-size key records by
 * hash prefix with an all-to-all, insert at the owner.)  Uniform-length
 * batches only: every key of the engine has the same word count. */

/* Words (8 bytes each) of the packed key of a record with these mate lengths. */
uint32_t fqd_key_words(uint32_t len0, uint32_t len1);

/* Encodes n uniform reads into n*(key_words+1) uint64 at `records` (device):
 * record i = [hash, key words...].  Does not touch the set (any engine of the right `segments` serves). */
int  fqd_encode_uniform(fqd_engine* e, const fqd_reads* seg, uint64_t n, uint64_t* records);

/* The same for reads of SEVERAL lengths (trimmed reads; any descriptors, ragged or uniform): record i = [hash,
 * len0 | len1 << 32, key words..., zeros] with every key padded to the width of the longest read the caller allows
 * (max_len0, max_len1): fqd_padded_key_words(max_len0, max_len1) words after the hash.  Equal padded keys <=> equal
 * sequences of equal lengths, so such records travel through the fixed-size exchange like any others; their owner
 * holds them as opaque keys: fqd_reserve_keys / fqd_insert_keys / fqd_insert_slabs with len0 = fqd_padded_key_words(..)
 * and len1 = FQD_OPAQUE_KEYS.  A read longer than the maxima is an error (FQD_ERR_ARG at the next fqd_engine_sync). */
#define FQD_OPAQUE_KEYS 0xFFFFFFFFu
uint32_t fqd_padded_key_words(uint32_t max_len0, uint32_t max_len1);
int  fqd_encode_padded(fqd_engine* e, const fqd_reads* seg, uint64_t n, uint32_t max_len0, uint32_t max_len1, uint64_t* records);

/* Stable partition of n records by owner = (hash >> 40) % n_parts: `out_keys` gets the KEY WORDS of the records
 * grouped by owner in input order (the hash is left out: 8 bytes less per record on the wire; the owner recomputes
 * it), `counts` (device, n_parts uint64) the group sizes, `origin` (device, n uint32) the input position of each
 * output key.  fqd_reserve_keys: *slot (device) is room for n keys at the tail of the engine's key store — let the
 * exchange write the keys there, then pass *slot to fqd_insert_keys, which inserts them where they lie, in order
 * (first-occurrence-wins follows the order of arrival); no other engine call may come in between.  The owner is an
 * ordinary uniform engine whose keys lie back to back, aligned like everywhere else. */
int  fqd_partition_keys(fqd_engine* e, const uint64_t* records, uint64_t n, uint32_t key_words,
                        uint32_t n_parts, uint64_t* out_keys, uint64_t* counts, uint32_t* origin);
int  fqd_reserve_keys(fqd_engine* e, uint64_t n, uint32_t len0, uint32_t len1, uint64_t** slot);
int  fqd_insert_keys(fqd_engine* e, const uint64_t* keys, uint64_t n, uint32_t len0, uint32_t len1, uint8_t* keep);

/* An owner whose key shape has to change in mid-run: the reference keys a read of any length at any point of the
 * file (seq_utils.cpp:35-49, hash_dup_remover.hpp:126-144), while the exchange moves keys of ONE width.  Every key the
 * engine holds (from fqd_insert_keys / fqd_insert_slabs, with known mate lengths or opaque) is laid out again as an
 * opaque key of new_words words — [len0 | len1 << 32][key words][zeros], the form fqd_encode_padded gives the same
 * read under maxima with fqd_padded_key_words(..) == new_words — record numbers unchanged, the set rebuilt from the
 * new keys.  Afterwards the engine takes keys with len0 = new_words, len1 = FQD_OPAQUE_KEYS.  Waits for the stream. */
int  fqd_widen_keys(fqd_engine* e, uint32_t new_words);

/* The exchange with messages of FIXED size, so that an all-to-all can be queued before anybody knows how many keys
 * go where (fqd_shard below).  fqd_partition_slabs is fqd_partition_keys with part p's keys written to slab p — the
 * slab_cap key slots from out_keys + p * slab_cap * key_words — and the keys a slab has no room for written, part
 * after part, behind the last slab (from slot n_parts * slab_cap; counts[p] is the true count, so counts[p] - slab_cap
 * of them spilled).  origin (n_parts * slab_cap + n entries) = input position per slot, 0xFFFFFFFF for a slab slot
 * nothing went to — fqd_scatter_flags passes over those.  fqd_insert_slabs inserts n_slabs slabs of slab_cap slots
 * lying back to back at the slot fqd_reserve_keys(n_slabs * slab_cap) gave, of which slab j holds
 * min(slab_count[j], slab_cap) keys (slab_count on the device); the unused slots become records that nothing is ever
 * compared with, so the owner's record numbering — and with it first-occurrence-wins — follows (slab, position).
 * keep has n_slabs * slab_cap entries; those of unused slots mean nothing. */
int  fqd_partition_slabs(fqd_engine* e, const uint64_t* records, uint64_t n, uint32_t key_words, uint32_t n_parts,
                         uint64_t slab_cap, uint64_t* out_keys, uint64_t* counts, uint32_t* origin);
/* Encode + group in one call, with every owner's slab CUT into n_chunks sub-slabs of sub_cap slots, one per chunk of
 * chunk_reads consecutive reads of the batch (n <= n_chunks * chunk_reads): chunk c's keys for part p go, in input order,
 * to the slots from (p * n_chunks + c) * sub_cap on.  Cutting the slab the way the input is cut is what lets the keys be
 * written ONCE, straight to their place, by an encoder whose workgroups need nothing from each other (csrc/fqd_kernels.hpp,
 * encode_chunks) — where that applies: reads of one fixed length per mate, at most 16 parts, chunk_reads a multiple of 256.
 * chunk_counts[p * n_chunks + c] and totals[p] (device; totals has n_parts + 1 words, the last says which way the slabs
 * were filled: 0 sub-slab by sub-slab, 1 from their first slot on) are the TRUE counts; a key whose sub-slab is full is not written:
 * call again with FQD_SLABS_EXACT for the three-step path (fqd_encode_uniform + fqd_partition_slabs through an internal
 * buffer), which fills each slab from its first slot on — the same thing seen as full, partial and empty sub-slabs — and
 * writes what n_chunks * sub_cap slots cannot take to the spill region (chunk_counts shows it as a count above sub_cap in
 * the last sub-slab).  origin[] as fqd_partition_slabs has it.  Owner side: fqd_insert_slabs with n_parts_of_the_group *
 * n_chunks slabs of sub_cap slots. */
#define FQD_SLABS_EXACT 1u
int  fqd_encode_slabs(fqd_engine* e, const fqd_reads* seg, uint64_t n, uint32_t n_parts, uint64_t chunk_reads, uint32_t n_chunks,
                      uint64_t sub_cap, uint64_t* out_keys, uint64_t* chunk_counts, uint64_t* totals, uint32_t* origin, uint32_t flags);
int  fqd_insert_slabs(fqd_engine* e, const uint64_t* keys, uint32_t n_slabs, uint64_t slab_cap, const uint64_t* slab_count,
                      uint32_t len0, uint32_t len1, uint8_t* keep);
/* The same two with every key's 8-byte placement hash beside it, so that the owner does not have to read all 64 bytes
 * of every arrived key once more just to hash it (1.75 ms per 100 M reads of a 14 ms sharded step, measured on one rank;
 * it costs the links 12.5 % more bytes): fqd_encode_slabs_hashed also writes hash[slot] for every key slot it fills
 * (one-pass grouping only: after FQD_SLABS_EXACT, or whenever totals[n_parts] reads 1, out_hashes holds nothing and the
 * owner uses fqd_insert_slabs); fqd_insert_slabs_hashed takes the hashes as they arrived (device, n_slabs * slab_cap
 * words, overwritten where a slot holds no key) — keys of known mate lengths only: an owner of opaque (padded) keys
 * hashes the padded words, which the source never saw. */
int  fqd_encode_slabs_hashed(fqd_engine* e, const fqd_reads* seg, uint64_t n, uint32_t n_parts, uint64_t chunk_reads, uint32_t n_chunks,
                             uint64_t sub_cap, uint64_t* out_keys, uint64_t* out_hashes, uint64_t* chunk_counts, uint64_t* totals,
                             uint32_t* origin, uint32_t flags);
int  fqd_insert_slabs_hashed(fqd_engine* e, const uint64_t* keys, uint64_t* hashes, uint32_t n_slabs, uint64_t slab_cap,
                             const uint64_t* slab_count, uint32_t len0, uint32_t len1, uint8_t* keep);

/* ---- one dedup job over the GPUs of a node (BASELINE north_star: "reads are partitioned across the 8
 * GPUs of one node by hash prefix with an RCCL all-to-all over xGMI so each GPU owns a disjoint bucket range") ----
 * The reference runs on one core and has nothing to mirror here.  A shard group is `world` ranks with one engine
 * (one GPU) each; a process hosts n_local consecutive ranks of it — all of them (the CLI's FQD_DEVICES run, tests
 * with several ranks on one card) or one (bench.py: one process per GPU under torch.distributed.run).  Global input
 * order is (round, rank, position).  Per round every rank encodes its reads, writes the keys bound for owner
 * d = (hash >> 40) % world into slab d of its send buffer (fqd_partition_slabs), ONE all-to-all of fixed-size slabs
 * moves them — queued before any count has reached a host — and every owner inserts what it received in (source
 * rank, position) order (fqd_insert_slabs); the flags return through the reverse all-to-all.  Rounds are pipelined:
 * the flags of a round are on their way once the NEXT fqd_shard_round (or fqd_shard_flush) has returned.  A slab
 * that overflows (one owner drawing far more than its share) is settled by a second, exactly sized exchange before
 * the owner inserts anything of that round: results are exact whatever the skew. */
typedef struct fqd_shard fqd_shard;
#define FQD_SHARD_ID_BYTES 128
#define FQD_SHARD_RCCL 0    /* ncclSend/ncclRecv in one group per exchange: a direct all-to-all over xGMI */
#define FQD_SHARD_COPY 1    /* peer copies; every rank must live in this process (always used when ranks share a GPU) */
#define FQD_SHARD_PADDED 1u /* fqd_shard_config.flags: reads of several lengths, exchanged as padded keys (fqd_encode_padded) */
#define FQD_SHARD_SEND_HASH 2u /* fqd_shard_config.flags: every key's placement hash travels with it (72 instead of 64 bytes per
                                  150-base read on the links; the owners skip their re-hash pass).  Ignored with FQD_SHARD_PADDED. */

typedef struct fqd_shard_config {
    int32_t  world;            /* ranks of the job                                                        */
    int32_t  n_local;          /* ranks this process hosts: engines[0 .. n_local)                          */
    int32_t  first_rank;       /* global rank of engines[0]                                                */
    int32_t  transport;        /* FQD_SHARD_RCCL / FQD_SHARD_COPY                                          */
    uint64_t round_reads;      /* most records (pairs) one rank brings to a round                          */
    uint32_t len0, len1;       /* the job's fixed read lengths (len1 = 0: single-end); with FQD_SHARD_PADDED the
                                  longest reads allowed: batches of any lengths up to them, ragged or uniform */
    uint32_t slack_permille;   /* slab capacity over a fair share, 0 = 30                                  */
    uint32_t flags;            /* FQD_SHARD_PADDED, FQD_SHARD_SEND_HASH                                    */
    uint64_t slab_records;     /* 0 = fqd_shard_slab_capacity(...); tests force overflows with a small one */
    const uint8_t* unique_id;  /* RCCL: FQD_SHARD_ID_BYTES from fqd_shard_unique_id, the same in every process */
} fqd_shard_config;

typedef struct fqd_shard_stats {
    uint64_t rounds, overflow_rounds;       /* rounds started; rounds in which a slab to or from this rank overflowed */
    uint64_t bytes_sent, bytes_received;    /* over the exchange, both directions, self included                      */
    uint64_t slab_records;                  /* key slots per slab                                                     */
    double   exchange_ms;                   /* device time of the forward all-to-alls (HIP events on its stream)      */
    int32_t  transport, ranks_in_comm;      /* what is in use; ncclCommCount of this rank's communicator              */
} fqd_shard_stats;

/* Rank 0 of a multi-process group makes the id; every process hands the same bytes to fqd_shard_create. */
int  fqd_shard_unique_id(uint8_t* id);
uint64_t fqd_shard_slab_capacity(uint64_t round_reads, int32_t world, uint32_t slack_permille);
int  fqd_shard_create(fqd_engine* const* engines, const fqd_shard_config* cfg, fqd_shard** out);
int  fqd_shard_destroy(fqd_shard* s);
/* One round.  seg: n_local * segments descriptors (uniform, device memory), n: reads per local rank (0 allowed: every
 * rank of the group must call once per round), keep: one device array per local rank.  Asynchronous. */
int  fqd_shard_round(fqd_shard* s, const fqd_reads* seg, const uint64_t* n, uint8_t* const* keep);
/* Completes every round in flight and waits; FQD_ERR_BAD_BASE when an engine met a byte outside {A,C,G,T,N}
 * (fqd_bad_base of that engine: `record` counts within the round that held it). */
int  fqd_shard_flush(fqd_shard* s);
/* Waits until keep[] of round `round` (0-based count of fqd_shard_round calls) is final.  FQD_ERR_BAD_BASE when a
 * local engine had met a byte outside {A,C,G,T,N} by the time it had encoded that round: fqd_shard_bad_base names
 * the first one in input order among this process's ranks (`record` counts within the rank's batch of the round;
 * flags of the records before it are valid). */
int  fqd_shard_wait(fqd_shard* s, uint64_t round);
int  fqd_shard_bad_base(fqd_shard* s, uint64_t round, int32_t* local_rank, uint64_t* record, uint32_t* segment,
                        uint32_t* position, uint8_t* byte);
int  fqd_shard_get_stats(fqd_shard* s, int32_t local_rank, fqd_shard_stats* out);
const char* fqd_shard_last_error(const fqd_shard* s);   /* NULL: last create() failure */

/* ---- the `--unordered` read-ID join (hash_dup_remover.hpp:150-192,257-347) ------------
 * The reference sorts both files by ID tag on disk (ExternalSorter<T>::sort,
 * external_sort.hpp:66-71,88-215, order = FastqViewWithId::cmp, fastqview.cpp:168-178) and
 * merge-joins the sorted files (hpp:279-340).  Here both files' tags are ordered together in
 * HBM by one hand-written radix sort over an order-preserving compact code of the tags, and the
 * matches are read off the sorted union (csrc/fqd_join.hip).  All pointers are device pointers. */

/* ID tags of one file: the tag bytes of record i are bytes[offsets[i] .. offsets[i]+lengths[i]) —
 * anywhere, e.g. inside the file's raw text as uploaded.  n < 2^31. */
typedef struct fqd_tags {
    const uint8_t*  bytes;
    const uint64_t* offsets;
    const uint32_t* lengths;
    uint64_t        n;
} fqd_tags;

/* FastqViewWithId::read_new / FastaViewWithId::read_new (fastqview.cpp:190-204,
 * fastaview.cpp:153-167) for n records whose ID lines lie in `text`: id_start[i] = offset of the
 * leading '@' / '>', id_len[i] = length of the ID line including its newline.  Writes the tag's
 * offset into `text` and its length: after the first '.' of the line if there is one, else after
 * the first byte; up to the first ' ', else through the end of the line including the newline. */
int  fqd_extract_tags(fqd_engine* e, const uint8_t* text, const uint64_t* id_start, const uint32_t* id_len,
                      uint64_t n, uint64_t* tag_off, uint32_t* tag_len);

/* perm[k] (n uint32) = index of the record with the k-th smallest tag in the order of
 * FastqViewWithId::cmp: bytewise over the shorter length, shorter first on a tie; equal tags keep
 * their input order.  Replaces ExternalSorter<T>::sort. */
int  fqd_sort_tags(fqd_engine* e, const fqd_tags* t, uint32_t* perm);

/* Result arrays of fqd_join_tags (device memory provided by the caller). */
typedef struct fqd_join {
    uint32_t* perm_a;    /* a->n: record of file 1 with the i-th smallest tag                    */
    uint32_t* perm_b;    /* b->n: the same for file 2                                            */
    uint32_t* match_a;   /* a->n: sorted position in file 2 of the partner of sorted record i,
                            0xFFFFFFFF when it has none                                          */
    uint32_t* match_b;   /* b->n: sorted position in file 1 of the partner, or 0xFFFFFFFF        */
    uint32_t* pair_a;    /* min(a->n, b->n): record of file 1 of the k-th pair, in tag order     */
    uint32_t* pair_b;    /* min(a->n, b->n): record of file 2 of the k-th pair                   */
    uint64_t* n_pairs;   /* HOST: number of pairs                                                */
} fqd_join;

/* The merge-join of two tag-sorted files (hpp:283-309) without the sorted files: record x of
 * file 1 and record y of file 2 are a pair iff their tags are equal and they have the same rank
 * among the records with that tag in their files (k-th with k-th, which is what the merge loop
 * does to repeated IDs).  This is the FULL inner join; the reference's end-of-file rule
 * (hpp:281,317-340) only ever drops the last pair and is applied by the caller from
 * perm/match.  Returns after the stream has drained (n_pairs is known). */
int  fqd_join_tags(fqd_engine* e, const fqd_tags* a, const fqd_tags* b, const fqd_join* out);

/* off_out[k] = off_table[idx[k]], len_out[k] = len_table[idx[k]] for k < n: turns a pair list into
 * the ragged sequence descriptors (fqd_reads.offsets / lengths) of a dedup batch in tag order. */
int  fqd_gather_seqs(fqd_engine* e, const uint32_t* idx, uint64_t n, const uint64_t* off_table,
                     const uint32_t* len_table, uint64_t* off_out, uint32_t* len_out);

/* ---- device halves of the bounded-memory `--unordered` run (host streams both files twice) ------
 * dst[dst_off[i] .. +len[i]) = src[src_off[i] .. +len[i]) for i < n: copies what the join and the
 * dedup need of every record (tag, sequence) out of an uploaded block into stores that stay in HBM. */
int  fqd_copy_spans(fqd_engine* e, const uint8_t* src, const uint64_t* src_off, const uint32_t* len, uint64_t n,
                    uint8_t* dst, const uint64_t* dst_off);

/* *count (host) = number of records of t whose tag is <= the tag of record other_index of `other`, in
 * the order of FastqViewWithId::cmp (fastqview.cpp:168-178): where the merge-join's other cursor
 * stands when a record without a partner is consumed (the end-of-file rule needs it, hpp:281,317-340). */
int  fqd_count_tags_le(fqd_engine* e, const fqd_tags* t, const fqd_tags* other, uint64_t other_index, uint64_t* count);

/* ---- `--unordered` over several GPUs: every GPU joins one contiguous RANGE of the tag order ------------------------
 * The reference's two sorted files are one global order (external_sort.hpp, FastqViewWithId::cmp); here it is cut at
 * n_split splitters (tags picked from a sample: fqd_sample_tags writes the tag of every (n / n_samples)-th record, cut
 * to `stride` bytes, into out_bytes + k * stride / out_len[k]; the host sorts them and picks).  fqd_classify_tags:
 * range_out[i] = number of splitters that are < tag i (splitter j = split_bytes + j * split_stride, split_len[j] bytes,
 * ascending) — records with equal tags always land in the same range.  fqd_range_keep: keep[i] = (range[i] == which),
 * *count (host) = how many: with fqd_output_plan (idx = NULL) and fqd_copy_spans that moves a range's whole records,
 * in input order, into one contiguous piece of text for the GPU that owns the range.  fqd_max_u32: the largest of n
 * values (the longest sequence: the key width of the pair exchange).  All device pointers unless said otherwise. */
int  fqd_sample_tags(fqd_engine* e, const fqd_tags* t, uint32_t n_samples, uint32_t stride, uint8_t* out_bytes, uint32_t* out_len);
int  fqd_classify_tags(fqd_engine* e, const fqd_tags* t, const uint8_t* split_bytes, uint32_t split_stride, const uint32_t* split_len,
                       uint32_t n_split, uint32_t* range_out);
int  fqd_range_keep(fqd_engine* e, const uint32_t* range, uint64_t n, uint32_t which, uint8_t* keep, uint64_t* count);
int  fqd_max_u32(fqd_engine* e, const uint32_t* values, uint64_t n, uint32_t* max_out);

/* Where the survivors go: pair k (tag order, k < n) is written iff keep[k]; dest[idx[k]] = sum of
 * sizes[idx[j]] over the kept pairs j < k (byte offset of record idx[k] in this file's output), entries
 * of records that are not written keep what the caller preset; *total (host) = output size in bytes.
 * Replaces the sorted temporary files of the reference as the way records reach their place
 * (external_sort.hpp:107-112,209-215). */
int  fqd_output_offsets(fqd_engine* e, const uint8_t* keep, const uint32_t* idx, uint64_t n, const uint32_t* sizes,
                        uint64_t* dest, uint64_t* total);

/* The same for a run whose text stays in HBM, per PAIR instead of per record: for pair k (tag order, k < n)
 * src_off[k] = starts[idx[k]] (where record idx[k] lies in the text), len[k] = sizes[idx[k]] if keep[k] else 0,
 * dst_off[k] = sum of len[j], j < k (where it goes in the output); *total (host) = output size.  idx == NULL:
 * pair k is record k (an ordered run: the output keeps the input's order).  A window
 * [a, b) of pairs is then assembled in output order by one fqd_copy_spans over src_off+a, len+a, dst_off+a. */
int  fqd_output_plan(fqd_engine* e, const uint8_t* keep, const uint32_t* idx, uint64_t n, const uint64_t* starts,
                     const uint32_t* sizes, uint64_t* src_off, uint32_t* len, uint64_t* dst_off, uint64_t* total);

/* `.gz` output made in HBM: the n bytes at src (device; text of FASTQ/FASTA records, a record = `lines_per_record`
 * lines: 4 or 2) become BGZF — gzip members (RFC 1952) of at most 65280 input bytes, each carrying its
 * compressed size in a 'BC' extra field — written back to back at dst (device, dst_capacity >=
 * fqd_bgzf_bound(n)); *out_bytes (host) = their total size.  The end-of-file marker member is the caller's to
 * append.  One deflate block per member: literals, byte runs and matches against the same column one record up,
 * under one pair of dynamic Huffman codes per call; a member that would not shrink is stored; CRC-32 computed on
 * the device.  Any inflater reads it; the ratio is about that of zlib level 1-2.  Returns when dst is complete.
 * Replaces the gzip compressor the reference pushes onto its output stream (file_utils.hpp:71-82) for outputs
 * whose bytes are already on the device. */
uint64_t fqd_bgzf_bound(uint64_t n);
int  fqd_bgzf_deflate(fqd_engine* e, const uint8_t* src, uint64_t n, uint32_t lines_per_record,
                      uint8_t* dst, uint64_t dst_capacity, uint64_t* out_bytes);

/* BGZF input inflated in HBM: member m (m < n_members; all arrays device) is the raw deflate stream of comp_len[m]
 * bytes at comp + comp_off[m] and inflates to exactly out_len[m] (<= 65536) bytes at text + out_off[m] whose CRC-32
 * is crc[m] — what the host read off the member's header and trailer.  One wave per member.  *n_bad (host) =
 * members whose stream is damaged, ends early or late, or whose CRC differs; text is then to be discarded.
 * Replaces the gzip decompressor the reference pushes onto its input stream (file_utils.hpp:58-69). */
int  fqd_bgzf_inflate(fqd_engine* e, const uint8_t* comp, const uint64_t* comp_off, const uint32_t* comp_len,
                      const uint64_t* out_off, const uint32_t* out_len, const uint32_t* crc, uint64_t n_members,
                      uint8_t* text, uint64_t* n_bad);
/* An ORDINARY gzip file (members that are one long deflate stream without member sizes: what gzip, pigz and sequencer software
 * write; the reference reads it through the same gzip decompressor, file_utils.cpp:59-66) inflated in HBM.  `deflate` (device;
 * any alignment; 32 readable bytes behind the last one) points at the FIRST member's raw deflate stream — the caller has walked
 * its 10-byte-plus header — and avail_bytes says how many bytes of the file lie from there on (trailers and further members
 * included).  The stream is cut into units whose block starts are GUESSED; every unit is decoded on its own — by the wave
 * decoder of the BGZF reader, into two texts behind two made-up 32 KiB windows, so that what comes out says for every byte whether it is a byte of
 * the stream or a copy of a place of the window before the unit — the chain of unit ends and starts is checked (a wrong guess:
 * the unit is decoded again from the true boundary), the windows are made unit after unit, and the places become text
 * (csrc/fqd_gunzip.hip).  Further members are walked where a final block ends; every member's CRC-32 and ISIZE are held against
 * its trailer.  *ok = 1: text[0 .. *text_bytes) is the text of all members, *deflate_bytes the offset (from `deflate`) at which
 * the LAST member's deflate stream ends (its 8-byte trailer is the end of the file's data), *crc32 that member's CRC-32.
 * *ok = 0: a guess that could not be repaired, damaged data, a CRC or length that is not the trailer's, bytes behind the last
 * member that are no member, a unit that outgrew its room, a text longer than text_cap, or more than 32 MiB of packed bytes in
 * which no dynamic block starts (stored or fixed blocks only: one wave's work) — nothing is reported beyond that: the
 * caller reads the file the host way, which produces the reference-visible diagnostic.  Waits for the stream. */
int  fqd_gunzip(fqd_engine* e, const uint8_t* deflate, uint64_t avail_bytes, uint8_t* text, uint64_t text_cap,
                uint64_t* text_bytes, uint64_t* deflate_bytes, uint32_t* crc32, int32_t* ok);
/* The same for a file that is STILL BEING COPIED to HBM by another thread of the caller: *arrived (host memory) says how many of
 * the avail_bytes bytes from `deflate` on are in HBM and complete — the caller raises it as its copies finish (after the copy
 * stream has been waited for), never lowers it, and ends at avail_bytes; ~0 means "the rest will not come" and ends the call
 * with *ok = 0.  Block starts are looked for and units decoded as their bytes arrive, so that most of the call runs under the
 * read of the file; the call returns when everything has arrived and been dealt with.  arrived = NULL: everything is there. */
int  fqd_gunzip_arriving(fqd_engine* e, const uint8_t* deflate, uint64_t avail_bytes, const volatile uint64_t* arrived,
                         uint8_t* text, uint64_t text_cap, uint64_t* text_bytes, uint64_t* deflate_bytes, uint32_t* crc32, int32_t* ok);

/* The same for one BATCH of members of a file that is still being read: queued on the engine's stream, nothing waited
 * for; bad members are ADDED to the two uint64 at bad_counters (device; zeroed by the caller, read when it likes), so
 * the inflate of what has arrived runs under the read of what has not. */
int  fqd_bgzf_inflate_async(fqd_engine* e, const uint8_t* comp, const uint64_t* comp_off, const uint32_t* comp_len,
                            const uint64_t* out_off, const uint32_t* out_len, const uint32_t* crc, uint64_t n_members,
                            uint8_t* text, uint64_t* bad_counters);

/* The n bytes of text (device) cut into records the way the reference's views do it (fastqview.cpp:92-138,
 * fastaview.cpp:78-100): a record is lines_per_record lines (4: FASTQ, 2: FASTA), starts with '@' / '>', and a FASTQ
 * record's sequence and quality lines are equally long.  fqd_count_lines gives the number of newlines; the caller
 * sizes the arrays for n_records = lines / lines_per_record and fqd_scan_records fills, per record, start (offset of
 * its first byte), seq_off (of its sequence line), id_len (ID line with its newline), seq_len (without), size (whole
 * record).  *well_formed (host) = 1 iff the text is exactly n_records such records, ending with its last byte;
 * otherwise the arrays are not to be used and the caller reads the file the host way, which reproduces the
 * reference's diagnostics for whatever is wrong. */
int  fqd_count_lines(fqd_engine* e, const uint8_t* text, uint64_t n, uint64_t* n_lines);
int  fqd_scan_records(fqd_engine* e, const uint8_t* text, uint64_t n, uint32_t lines_per_record, uint64_t n_records,
                      uint64_t* start, uint64_t* seq_off, uint32_t* id_len, uint32_t* seq_len, uint32_t* size, int* well_formed);

/* keep_out[origin[k]] = flags[k] for k < n: puts the flags that came back from the
 * owners (in partition order) into input order.  All device pointers. */
int  fqd_scatter_flags(fqd_engine* e, const uint8_t* flags, const uint32_t* origin, uint64_t n, uint8_t* keep_out);

/* ---- synthetic workload (bench.py, tests): SURVEY §8(d) ---------------------
 * Fills `bases` (device) with n reads of `len` bases at stride `len`, global
 * indices [first, first+n): read g>0 is with probability dup_permille/1000 a
 * copy of a uniformly chosen earlier read (chains allowed), else fresh uniform
 * ACGT (one N in about 1 read of 8).  mate = 0/1 selects the mate stream of a
 * pair; for mate 1 a copied pair keeps its parent's mate-2 with probability 1/2.
 * expect_keep (device, may be NULL) receives the analytically known flags. */
int  fqd_synth_reads(fqd_engine* e, uint64_t seed, uint64_t first, uint64_t n, uint32_t len,
                     uint32_t dup_permille, int mate, uint8_t* bases, uint8_t* expect_keep);

#ifdef __cplusplus
}
#endif
#endif /* FQDUPAWAY_H */
