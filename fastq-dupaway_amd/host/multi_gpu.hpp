// multi_gpu.hpp — which GPUs a run uses.  The multi-GPU run itself is HashDupRemover::run_ordered_multi over the shard
// group of include/fqdupaway.h (fqd_shard_*, csrc/fqd_shard.hip); the reference is one thread on one core and has
// nothing to mirror here.
#pragma once
#include <vector>

namespace fqdhost {

// FQD_DEVICES="0,1,2,3" -> {0,1,2,3}; empty when unset.  A device may repeat (several ranks on one card: rehearsals).
std::vector<int> devices_from_env();

} // namespace fqdhost
